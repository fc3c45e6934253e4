"""Inverted-residual block micro-benchmark (HIP events, kernels alone on the chip).
forward: unfused chain  pq_conv1x1_fwd (expand, stores E channels + statistics) -> pq_dw3x3_fwd
         fused          pq_conv1x1_fwd(out = NULL: statistics only) -> pq_xdw_fwd
at the shapes of the mobilenetv2-fpn 512 x 512 batch-16 step."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
SHAPES = [(16, 256, 256, 16, 96, 2), (16, 128, 128, 24, 144, 1), (16, 128, 128, 24, 144, 2), (16, 64, 64, 32, 192, 1),
          (16, 64, 64, 32, 192, 2)]
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'
def vec(n): return torch.rand(n + 32, device=dev) + 0.5
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0] * 5
for (B, H, W, I, E, s) in SHAPES:
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * H * W
    x = torch.randn(B, H, W, I, device=dev); we = torch.randn(E, I, device=dev) * 0.2; wd = torch.randn(E, 9, device=dev)
    ye = torch.empty(B, H, W, E, device=dev); out = torch.empty(B, OH, OW, E, device=dev)
    xa, xb, ma, mb = vec(I), vec(I), vec(E), vec(E)
    nb1 = l.pq_conv1x1_blocks_k(M, E, I, 0)
    p1 = torch.empty(nb1 * 2 * E, device=dev)
    f = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), xa.data_ptr(), xb.data_ptr(), I, 0), we.data_ptr(), None, ye.data_ptr(), p1.data_ptr(),
                       M, I, E, E, E, 1, 0, 0, 0, 0)
    t_e = timeit(lambda: abi.check(l.pq_conv1x1_fwd(C.byref(f), st)))
    nb2 = l.pq_dw3x3_fwd_blocks(B, OH, OW, E, s)
    p2 = torch.empty(nb2 * 2 * E, device=dev)
    d = abi.DwFwd(abi.Src(ye.data_ptr(), ma.data_ptr(), mb.data_ptr(), E, 2), wd.data_ptr(), out.data_ptr(), p2.data_ptr(), B, H, W, OH, OW, E, s, E, E)
    t_d = timeit(lambda: abi.check(l.pq_dw3x3_fwd(C.byref(d), st)))
    f0 = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), xa.data_ptr(), xb.data_ptr(), I, 0), we.data_ptr(), None, None, p1.data_ptr(),
                        M, I, E, 0, E, 1, 0, 0, 0, 0)
    t_0 = timeit(lambda: abi.check(l.pq_conv1x1_fwd(C.byref(f0), st)))
    sc = torch.zeros(l.pq_xstat_fwd_scratch_bytes(M, I) // 4, device=dev)
    o4 = [vec(E) for _ in range(4)]
    xs = abi.XstatFwd()
    xs.inp = f.inp
    xs.we, xs.gamma, xs.beta = we.data_ptr(), ma.data_ptr(), mb.data_ptr()
    xs.mean, xs.invstd, xs.scale, xs.shift = [v.data_ptr() for v in o4]
    xs.scratch, xs.M, xs.I, xs.E, xs.eps, xs.momentum = sc.data_ptr(), M, I, E, 1e-5, 0.1
    t_g = timeit(lambda: abi.check(l.pq_xstat_fwd(C.byref(xs), st)))
    nb3 = l.pq_xdw_fwd_blocks(B, H, W, I, E, s)
    p3 = torch.empty(nb3 * 2 * E, device=dev)
    a = abi.XdwFwd()
    a.inp = f.inp
    a.we, a.wd = we.data_ptr(), wd.data_ptr()
    a.mid = abi.Src(None, ma.data_ptr(), mb.data_ptr(), 0, 2)
    a.out, a.partial = out.data_ptr(), p3.data_ptr()
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, H, W, OH, OW, I, E, s
    a.ldy, a.ldo, a.ldp = E, E, E
    t_x = timeit(lambda: abi.check(l.pq_xdw_fwd(C.byref(a), st)))
    byt = 4.0 * (M * I + B * OH * OW * E)
    fl = 2.0 * M * I * E
    for k, v in enumerate((t_e, t_d, t_0, t_x, t_g)): tot[k] += v
    print('B=%d %dx%d %d->%d s=%d | unfused: expand %.3f + dw %.3f = %.3f ms | fused: moments %.3f (statistics-only GEMM %.3f) + xdw %.3f = %.3f ms '
          '(xdw: %.0f GB/s of minimal traffic, %.1f TF/s of the expand, %d walkers)' % (
              B, H, W, I, E, s, t_e, t_d, t_e + t_d, t_g, t_0, t_x, t_g + t_x, byt / t_x / 1e6, fl / t_x / 1e9, nb3), flush=True)
print('sum: unfused %.3f + %.3f = %.3f ms | fused %.3f + %.3f = %.3f ms (statistics-only GEMM instead of the moments pass: %.3f)' % (
    tot[0], tot[1], tot[0] + tot[1], tot[4], tot[3], tot[4] + tot[3], tot[2]))
