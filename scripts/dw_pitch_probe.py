"""Does the channel PITCH matter to the streaming depthwise kernels?  pq_dw3x3_fwd / pq_dw3x3_dgrad on 16 x 128 x 128 planes with
C channels stored at pitch ld >= C (C = 144 at ld 144 / 160 / 192, and the aligned neighbours C = 128, 160 for reference)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'
def vec(n): return torch.rand(n + 32, device=dev) + 0.5
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
s = 1
for B, H, W, Cc, ld in ((16, 128, 128, 128, 128), (16, 128, 128, 144, 144), (16, 128, 128, 144, 160), (16, 128, 128, 144, 192), (16, 128, 128, 160, 160),
                        (16, 128, 128, 32, 32), (16, 256, 256, 32, 32), (4, 256, 256, 128, 128), (64, 128, 128, 32, 32), (16, 128, 128, 64, 64), (16, 256, 256, 64, 64)):
    x = torch.randn(B, H, W, ld, device=dev); out = torch.empty(B, H, W, ld, device=dev)
    yo = torch.randn(B, H, W, ld, device=dev); g = torch.randn(B, H, W, ld, device=dev); gin = torch.empty(B, H, W, ld, device=dev)
    w = torch.randn(Cc, 9, device=dev)
    a, b2, mn, iv, p, q, mu, r = (vec(ld) for _ in range(8))
    nb = max(l.pq_dw3x3_fwd_blocks(B, H, W, Cc, s), l.pq_ew_blocks(B * H * W, Cc))
    ldp = (Cc + 15) // 16 * 16
    part = torch.empty(nb * 2 * ldp, device=dev)
    f = abi.DwFwd(abi.Src(x.data_ptr(), a.data_ptr(), b2.data_ptr(), ld, 2), w.data_ptr(), out.data_ptr(), part.data_ptr(), B, H, W, H, W, Cc, s, ld, ldp, 0)
    gs = abi.GSrc(g.data_ptr(), yo.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), ld, ld)
    sink = abi.GSink(gin.data_ptr(), x.data_ptr(), a.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), ld, ld, 2, 0, 1, ldp)
    d = abi.DwDgrad(gs, w.data_ptr(), sink, B, H, W, H, W, Cc, s)
    tf = timeit(lambda: abi.check(l.pq_dw3x3_fwd(C.byref(f), st)))
    td = timeit(lambda: abi.check(l.pq_dw3x3_dgrad(C.byref(d), st)))
    n = B * H * W * Cc * 4
    print('B=%2d %3dx%3d C=%3d ld=%3d | fwd %6.1f us %5.0f GB/s useful | dgrad %6.1f us %5.0f GB/s useful' % (B, H, W, Cc, ld, tf * 1e3, 2 * n / tf / 1e6, td * 1e3, 4 * n / td / 1e6))
