"""BN finalize micro-benchmark (HIP events, back-to-back launches)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'
for (nblk, Cc) in [(32, 960), (32, 160), (128, 384), (512, 192), (2048, 144), (8192, 96), (8192, 16)]:
    ldp = (Cc + 15) // 16 * 16
    part = torch.rand(nblk * 2 * ldp, device=dev)
    g, b, rm, rv, mean, inv, sc, sh = (torch.rand(Cc + 32, device=dev) + 0.5 for _ in range(8))
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    scr = torch.empty(l.pq_bn_finalize_scratch_bytes(Cc) // 8, dtype=torch.float64, device=dev)
    tk = torch.zeros(l.pq_bn_finalize_tickets(Cc), dtype=torch.int32, device=dev)
    a = abi.BnFwdFin(part.data_ptr(), g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                     sc.data_ptr(), sh.data_ptr(), nblk, ldp, Cc, 1, float(nblk * 128), 1e-5, 0.1, scr.data_ptr(), tk.data_ptr())
    for _ in range(5): abi.check(l.pq_bn_fwd_finalize(C.byref(a), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): abi.check(l.pq_bn_fwd_finalize(C.byref(a), st))
    e1.record(); torch.cuda.synchronize()
    print('nblk %5d C %4d : %.2f us per launch (back-to-back)' % (nblk, Cc, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
# reference: trivial fill kernel
x = torch.empty(1024, device=dev)
for _ in range(5): l.pq_fill(x.data_ptr(), 0.0, 1024, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200): l.pq_fill(x.data_ptr(), 0.0, 1024, st)
e1.record(); torch.cuda.synchronize()
print('fill 1024 floats: %.2f us per launch' % (e0.elapsed_time(e1) / 200 * 1e3))
