"""Which main-stream ops of the backward pay for the weight-gradient stream: per op, HIP-event time in the two-stream replay
(pq_run_ops2_timed, as in the step) against the one-stream replay (pq_run_ops_timed: nothing runs beside it).
usage: [MODEL=...] python scripts/op_inflation.py <batch> <size> [top]"""
import ctypes as C, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pqdet_amd import abi
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel

B, S = int(sys.argv[1]), int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
model = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('MODEL', 'mobilenetv2-fpn')))).cuda(); model.train()
x = torch.randn(B, 3, S, S).cuda()
tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, S, 20, seed=1))
plan = model._plan(x, True)
plan._patch(x, tgt)
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = C.c_void_p(model._side_stream().cuda_stream)
abi.check(l.pq_run_ops2(plan.fwd_arr, len(plan.fwd), st, sd))
arr, n = plan.bwd_arr, len(plan.bwd)
def timed(two, reps=8):
    ms = (C.c_float * n)(); tot = [0.0] * n
    for _ in range(reps):
        abi.check(l.pq_run_ops2_timed(arr, n, st, sd, ms) if two else l.pq_run_ops_timed(arr, n, st, ms))
        for k in range(n): tot[k] += ms[k] / reps
    return tot
one, two = timed(False), timed(True)
rows = []
for k in range(n):
    op = arr[k]
    if op.code == abi.OP_FILL and op.u.fill.n == 0: continue
    side = bool(op.flags & abi.OPF_SIDE)
    name = bench.KERNEL_CLASS.get(op.code, ('other',))[0]
    a = getattr(op.u, abi.OP_FIELD[op.code])
    shape = ''
    if op.code in (abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD, abi.OP_CONV1X1_BWD): shape = 'M%d I%d O%d' % (a.M, a.I, a.O)
    elif op.code == abi.OP_DW_BWD: shape = 'B%d %dx%d C%d s%d' % (a.B, a.IH, a.IW, a.C, a.stride)
    rows.append((two[k] - one[k], k, 'side' if side else 'main', name[:34], shape, one[k], two[k]))
main = [r for r in rows if r[2] == 'main']
print('main-stream ops: alone %.3f ms, in the two-stream replay %.3f ms; side-stream ops: %.3f -> %.3f ms' % (
    sum(r[5] for r in main), sum(r[6] for r in main), sum(r[5] for r in rows if r[2] == 'side'), sum(r[6] for r in rows if r[2] == 'side')))
print('largest inflations (main stream):')
for d, k, w, name, shape, a1, a2 in sorted(main, reverse=True)[:top]:
    print('  op %3d %-34s %-22s alone %6.1f us  in step %6.1f us  (+%5.1f)' % (k, name, shape, a1 * 1e3, a2 * 1e3, d * 1e3))
print('in op order (main | side):')
for d, k, w, name, shape, a1, a2 in rows:
    print('  %3d %s %-34s %-22s %6.1f -> %6.1f' % (k, w, name, shape, a1 * 1e3, a2 * 1e3))
