"""Floor of a dependent kernel launch on one stream: N pq_fill ops replayed by ONE pq_run_ops call (the executor's C++ loop,
as in a train step), HIP events around the call."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
buf = torch.zeros(1 << 22, device='cuda')
N = 500
def run(n):
    ops = (abi.Op * N)()
    for k in range(N):
        ops[k].code = abi.OP_FILL
        ops[k].u.fill.dst = buf.data_ptr(); ops[k].u.fill.value = 1.0; ops[k].u.fill.n = n
    for _ in range(3): abi.check(l.pq_run_ops(ops, N, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4): abi.check(l.pq_run_ops(ops, N, st))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * N) * 1e3
for n in (4, 1024, 65536, 1 << 20, 1 << 22):
    print('fill of %8d floats: %.2f us per launch (%d back to back, one pq_run_ops call)' % (n, run(n), N))
