"""Grouped 3x3 conv micro-benchmark: pq_gconv3x3_{fwd,dgrad,wgrad} back to back on regnetx layer shapes.
usage: gconv_bench.py [fwd,dgrad,wgrad] [reps]   (SHAPES="B,H,W,C,gw,s;..." overrides the list)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi

SHAPES = [(16, 256, 256, 48, 24, 2), (16, 128, 128, 96, 24, 2), (16, 64, 64, 96, 24, 1), (16, 64, 64, 240, 24, 2),
          (16, 32, 32, 240, 24, 1), (16, 32, 32, 528, 24, 2), (16, 16, 16, 528, 24, 1), (16, 64, 64, 80, 16, 1),
          (16, 32, 32, 176, 16, 1)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd,dgrad,wgrad'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'

def vec(n): return torch.rand(n + 32, device=dev) + 0.5

def timeit(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for (B, H, W, Cc, gw, s) in SHAPES:
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cc, device=dev); y = torch.randn(B, OH, OW, Cc, device=dev); g = torch.randn(B, OH, OW, Cc, device=dev)
    w = torch.randn(Cc, gw, 3, 3, device=dev) * 0.1; gin = torch.empty(B, H, W, Cc, device=dev)
    sc, sh, p, q, mu, r, mn, iv = [vec(Cc) for _ in range(8)]
    ldp = (Cc + 15) // 16 * 16
    flops = 2.0 * B * OH * OW * Cc * gw * 9
    line = 'B=%d %dx%d C=%d gw=%d s=%d ' % (B, H, W, Cc, gw, s)
    src = abi.Src(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), Cc, 1)
    dy = abi.GSrc(g.data_ptr(), y.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), Cc, Cc)
    if 'fwd' in which:
        part = torch.empty(l.pq_gconv3x3_blocks(B, OH, OW) * 2 * ldp, device=dev)
        a = abi.GconvFwd(src, w.data_ptr(), y.data_ptr(), part.data_ptr(), B, H, W, OH, OW, Cc, gw, s, Cc, ldp)
        ms = timeit(lambda: abi.check(l.pq_gconv3x3_fwd(C.byref(a), st)))
        line += '| fwd %.3f ms %5.1f TF ' % (ms, flops / ms / 1e9)
    if 'dgrad' in which:
        part = torch.empty(l.pq_gconv3x3_blocks(B, H, W) * 2 * ldp, device=dev)
        d = abi.GconvDgrad()
        d.dy, d.w = dy, w.data_ptr()
        d.dx = abi.GSink(gin.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), Cc, Cc, 1, 0, 1, ldp)
        d.B, d.IH, d.IW, d.OH, d.OW, d.C, d.gw, d.stride = B, H, W, OH, OW, Cc, gw, s
        ms = timeit(lambda: abi.check(l.pq_gconv3x3_dgrad(C.byref(d), st)))
        line += '| dgrad %.3f ms %5.1f TF ' % (ms, flops / ms / 1e9)
    if 'wgrad' in which:
        scratch = torch.empty(l.pq_gconv3x3_wgrad_scratch_bytes(B, OH, OW, Cc, gw) // 4 + 16, device=dev)
        dw = torch.empty(Cc, gw, 3, 3, device=dev)
        wg = abi.GconvWgrad()
        wg.dy, wg.inp, wg.dw, wg.scratch = dy, src, dw.data_ptr(), scratch.data_ptr()
        wg.B, wg.IH, wg.IW, wg.OH, wg.OW, wg.C, wg.gw, wg.stride = B, H, W, OH, OW, Cc, gw, s
        ms = timeit(lambda: abi.check(l.pq_gconv3x3_wgrad(C.byref(wg), st)))
        line += '| wgrad %.3f ms %5.1f TF' % (ms, flops / ms / 1e9)
    print(line, flush=True)
