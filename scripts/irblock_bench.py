"""Inverted-residual block micro-benchmark (HIP events, kernels alone on the chip).
forward: unfused chain  pq_conv1x1_fwd (expand, stores E channels + statistics) -> pq_dw3x3_fwd
         fused          pq_conv1x1_fwd(out = NULL: statistics only) -> pq_xdw_fwd
at the shapes of the mobilenetv2-fpn 512 x 512 batch-16 step."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
SHAPES = [(16, 256, 256, 16, 96, 2), (16, 128, 128, 24, 144, 1), (16, 128, 128, 24, 144, 2)]
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
for (B, H, W, I, E, s) in ([] if os.environ.get('BWD_ONLY') else SHAPES):
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

# ---- backward: walking depthwise backward + fused narrow 1x1 backward (expand layer)  vs  pq_xdw_bwd
print('backward')
totb = [0.0] * 3
for (B, H, W, I, E, s) in SHAPES:
    if not l.pq_xdw_bwd_blocks(B, H, W, I, E, s):
        continue
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * H * W
    x = torch.randn(B, H, W, I, device=dev); we = torch.randn(E, I, device=dev) * 0.2; wd = torch.randn(E, 9, device=dev)
    ye = torch.randn(B, H, W, E, device=dev); ge = torch.empty(B, H, W, E, device=dev)
    yd = torch.randn(B, OH, OW, E, device=dev); gd = torch.randn(B, OH, OW, E, device=dev); gx = torch.empty(B, H, W, I, device=dev)
    vI = [vec(I) for _ in range(4)]; vE = [vec(E) for _ in range(13)]
    # unfused: pq_dw3x3_bwd part 5 (+6) then pq_conv1x1_bwd
    nbw = l.pq_dw3x3_bwd_walk_blocks(B, H, W, E, s)
    pw_ = torch.empty(nbw * 2 * E, device=dev); scw = torch.empty(nbw * E * 9 + 16, device=dev); dwd = torch.empty(E * 9, device=dev)
    gsd = abi.GSrc(gd.data_ptr(), yd.data_ptr(), vE[0].data_ptr(), vE[1].data_ptr(), vE[2].data_ptr(), vE[3].data_ptr(), E, E)
    srce = abi.Src(ye.data_ptr(), vE[4].data_ptr(), vE[5].data_ptr(), E, 2)
    sinke = abi.GSink(ge.data_ptr(), ye.data_ptr(), vE[4].data_ptr(), vE[5].data_ptr(), vE[6].data_ptr(), vE[7].data_ptr(), pw_.data_ptr(), E, E, 2, 0, 1, E)
    f5 = abi.DwBwd(gsd, srce, wd.data_ptr(), sinke, None, scw.data_ptr(), B, H, W, OH, OW, E, s, 5)
    f6 = abi.DwBwd(abi.GSrc(), abi.Src(), None, abi.GSink(), dwd.data_ptr(), scw.data_ptr(), B, H, W, OH, OW, E, s, 6)
    t_dw = timeit(lambda: (abi.check(l.pq_dw3x3_bwd(C.byref(f5), st)), abi.check(l.pq_dw3x3_bwd(C.byref(f6), st))))
    nb1 = l.pq_conv1x1_bwd_blocks(M, I, E)
    t_pw = float('nan')
    if nb1:
        p1 = torch.empty(nb1 * 2 * 32, device=dev); dwe = torch.empty(E * I, device=dev)
        sc1 = torch.empty(l.pq_conv1x1_bwd_scratch_bytes(M, I, E) // 4 + 16, device=dev)
        gse = abi.GSrc(ge.data_ptr(), ye.data_ptr(), vE[8].data_ptr(), vE[9].data_ptr(), vE[10].data_ptr(), vE[11].data_ptr(), E, E)
        sinkx = abi.GSink(gx.data_ptr(), x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), vI[2].data_ptr(), vI[3].data_ptr(), p1.data_ptr(), I, I, 0, 0, 1, 32)
        c1 = abi.Conv1x1Bwd(gse, we.data_ptr(), abi.Src(x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), I, 0), sinkx, dwe.data_ptr(), sc1.data_ptr(), M, I, E)
        t_pw = timeit(lambda: abi.check(l.pq_conv1x1_bwd(C.byref(c1), st)))
    # fused
    nfix = l.pq_xdw_bwd_fix_blocks(M, I)
    p2 = torch.empty(nfix * 2 * 32, device=dev)
    scr = torch.empty(l.pq_xdw_bwd_scratch_bytes(B, H, W, I, E, s) // 4, device=dev)
    mom = torch.rand(I * I + I, device=dev, dtype=torch.float64)
    outs = [torch.empty(E * I, device=dev), torch.empty(E * 9, device=dev), torch.empty(E, device=dev), torch.empty(E, device=dev)]
    a = abi.XdwBwd()
    a.inp = abi.Src(x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), I, 0)
    a.we, a.wd = we.data_ptr(), wd.data_ptr()
    a.mid_scale, a.mid_shift, a.mid_mean, a.mid_invstd, a.mid_gamma = [v.data_ptr() for v in vE[4:9]]
    a.mid_act = 2
    a.dy = gsd
    a.dx = abi.GSink(gx.data_ptr(), x.data_ptr(), vI[0].data_ptr(), vI[1].data_ptr(), vI[2].data_ptr(), vI[3].data_ptr(), p2.data_ptr(), I, I, 0, 0, 1, 32)
    a.dwe, a.dwd, a.dgamma, a.dbeta = [o.data_ptr() for o in outs]
    a.moments, a.scratch = mom.data_ptr(), scr.data_ptr()
    a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, H, W, OH, OW, I, E, s
    t_q = timeit(lambda: abi.check(l.pq_xdw_bwd(C.byref(a), st)))
    byt = 4.0 * (2 * B * OH * OW * E + 3 * M * I)
    totb[0] += t_dw; totb[1] += t_pw; totb[2] += t_q
    print('B=%d %dx%d %d->%d s=%d | unfused: depthwise backward %.3f + narrow 1x1 backward %.3f = %.3f ms | fused (3 launches) %.3f ms '
          '(%.0f GB/s of minimal traffic, %d walkers)' % (B, H, W, I, E, s, t_dw, t_pw, t_dw + t_pw, t_q, byt / t_q / 1e6,
                                                          l.pq_xdw_bwd_blocks(B, H, W, I, E, s)), flush=True)
print('sum backward: unfused %.3f + %.3f = %.3f ms | fused %.3f ms' % (totb[0], totb[1], totb[0] + totb[1], totb[2]))
