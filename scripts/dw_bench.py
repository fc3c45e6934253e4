"""Depthwise backward micro-benchmark: separate dgrad + wgrad kernels vs the fused kernel (HIP events)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
SHAPES = [(16, 256, 256, 32, 1), (16, 128, 128, 144, 1), (16, 64, 64, 192, 1), (16, 32, 32, 384, 1), (16, 32, 32, 576, 1),
          (16, 16, 16, 960, 1), (16, 64, 64, 128, 1), (16, 256, 256, 96, 2), (16, 128, 128, 144, 2)]
if os.environ.get('SHAPES'):          # e.g. SHAPES='16,16,16,512,1;16,32,32,256,1'
    SHAPES = [tuple(int(v) for v in it.split(',')) for it in os.environ['SHAPES'].split(';')]
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'
def vec(n): return torch.rand(n + 32, device=dev) + 0.5
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0, 0, 0, 0, 0]
for (B, H, W, Cc, s) in SHAPES:
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cc, device=dev); yo = torch.randn(B, OH, OW, Cc, device=dev); g = torch.randn(B, OH, OW, Cc, device=dev)
    w = torch.randn(Cc, 9, device=dev); gin = torch.empty(B, H, W, Cc, device=dev); dw = torch.empty(Cc, 9, device=dev)
    a, b2, mn, iv, p, q, mu, r = (vec(Cc) for _ in range(8))
    nb = max(l.pq_ew_blocks(B * H * W, Cc), l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, s))
    part = torch.empty(nb * 2 * Cc, device=dev)
    gs = abi.GSrc(g.data_ptr(), yo.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), Cc, Cc)
    src = abi.Src(x.data_ptr(), a.data_ptr(), b2.data_ptr(), Cc, 2)
    sink = abi.GSink(gin.data_ptr(), x.data_ptr(), a.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), Cc, Cc, 2, 0, 1, Cc)
    sc = torch.empty(max(l.pq_dw3x3_wgrad_scratch_bytes(B, OH, OW, Cc), l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, s) * Cc * 9 * 4) // 4 + 16, device=dev)
    d = abi.DwDgrad(gs, w.data_ptr(), sink, B, H, W, OH, OW, Cc, s)
    wg = abi.DwWgrad(gs, src, dw.data_ptr(), sc.data_ptr(), B, H, W, OH, OW, Cc, s)
    t1 = timeit(lambda: abi.check(l.pq_dw3x3_dgrad(C.byref(d), st)))
    t2 = timeit(lambda: abi.check(l.pq_dw3x3_wgrad(C.byref(wg), st)))
    t3 = 0.0
    nbw = l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, s)
    partw = torch.empty(nbw * 2 * Cc, device=dev); scw = torch.empty(nbw * Cc * 9 + 16, device=dev)
    sinkw = abi.GSink(gin.data_ptr(), x.data_ptr(), a.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), partw.data_ptr(), Cc, Cc, 2, 0, 1, Cc)
    f5 = abi.DwBwd(gs, src, w.data_ptr(), sinkw, None, scw.data_ptr(), B, H, W, OH, OW, Cc, s, 5)
    f6 = abi.DwBwd(abi.GSrc(), abi.Src(), None, abi.GSink(), dw.data_ptr(), scw.data_ptr(), B, H, W, OH, OW, Cc, s, 6)
    t5 = timeit(lambda: abi.check(l.pq_dw3x3_bwd(C.byref(f5), st)))
    t6 = timeit(lambda: abi.check(l.pq_dw3x3_bwd(C.byref(f6), st)))
    byt = 4.0 * (B * H * W * Cc * 2 + B * OH * OW * Cc * 2)     # minimal: g, y_out, y_in read + g_in written
    tot[0] += t1; tot[1] += t2; tot[2] += t3; tot[3] += t5; tot[4] += t6
    print('B=%d %dx%d C=%d s=%d | dgrad %.3f wgrad %.3f sum %.3f |%s walking %.3f ms (%.0f GB/s of minimal traffic, %d workgroup rows) + reduce %.3f' % (
        B, H, W, Cc, s, t1, t2, t1 + t2, '', t5, byt / t5 / 1e6, nbw, t6), flush=True)
print('sum dgrad %.3f wgrad %.3f fused tiled %.3f walking %.3f + reduce %.3f' % tuple(tot))
if os.environ.get('BWD_ONLY'): sys.exit(0)

# ---- forward
print('forward')
totf = 0
for (B, H, W, Cc, s) in SHAPES:
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cc, device=dev); out = torch.empty(B, OH, OW, Cc, device=dev); w = torch.randn(Cc, 9, device=dev)
    a, b2 = vec(Cc), vec(Cc)
    nb = l.pq_dw3x3_fwd_blocks(B, OH, OW, Cc, s)
    part = torch.empty(nb * 2 * Cc, device=dev)
    f = abi.DwFwd(abi.Src(x.data_ptr(), a.data_ptr(), b2.data_ptr(), Cc, 2), w.data_ptr(), out.data_ptr(), part.data_ptr(), B, H, W, OH, OW, Cc, s, Cc, Cc)
    tf = timeit(lambda: abi.check(l.pq_dw3x3_fwd(C.byref(f), st)))
    totf += tf
    print('B=%d %dx%d C=%d s=%d | fwd %.3f ms (%.0f GB/s)' % (B, H, W, Cc, s, tf, 4.0 * (B * H * W * Cc + B * OH * OW * Cc) / tf / 1e6), flush=True)
print('sum fwd %.3f' % totf)
