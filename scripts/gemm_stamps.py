"""Where a ring-GEMM launch spends its time: per-workgroup s_memtime stamps of the DIAGNOSTIC build
(python -m pqdet_amd.build -DPQ_STAMPS --tag=stamps; PQDET_HIP_LIB=pqdet_amd/lib/libpqdet_hip_stamps.so python scripts/gemm_stamps.py).
Prints, per shape and pass: workgroups, in-kernel clock, median / p90 cycles of the three phases (entry -> first k-tile landed,
main loop, epilogue), the span of the launch on the 100 MHz clock and how the workgroup starts are spread over it."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi

SHAPES = [(4096, 512, 1024), (4096, 1024, 512), (16384, 256, 512), (16384, 512, 256), (65536, 128, 256), (65536, 256, 128),
          (4096, 160, 960), (4096, 960, 160), (16384, 96, 576), (16384, 576, 96), (16384, 384, 64), (4096, 320, 1280), (4096, 1280, 512)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd,dgrad'
l = abi.lib()
l.pq_debug_set_stamps.restype = C.c_int
l.pq_debug_set_stamps.argtypes = [C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = 'cuda'
NWG = 1 << 16
stamps = torch.zeros(NWG * 8, dtype=torch.int64, device=dev)
assert l.pq_debug_set_stamps(stamps.data_ptr()) == 0


def vec(n):
    return torch.rand(n + 32, device=dev) + 0.5


def report(tag, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) * 1e3
    s = stamps.view(NWG, 8).cpu()
    used = s[:, 0] != 0
    s = s[used]
    n = s.shape[0]
    t0, t1, t2, t3, r0, r1 = (s[:, i].double() for i in range(6))
    pro, main, epi = t1 - t0, t2 - t1, t3 - t2
    span = (r1.max() - r0.min()).item() / 100.0            # us (100 MHz)
    clk = ((t3 - t0) / (r1 - r0).clamp(min=1)).median().item() * 100.0      # MHz
    starts = (r0 - r0.min()) / 100.0
    late = (starts > (r1 - r0.min()).min() / 100.0).sum().item()          # workgroups that started after the first one ended
    def q(v, f):
        return v.kthvalue(max(1, int(f * n)))[0].item()
    us = lambda cyc: cyc / clk
    t6, t7 = s[:, 6].double(), s[:, 7].double()
    extra = ''
    if (t6 > 0).all():
        extra = ' | epilogue split: loads landed +%.0f, blocks stored +%.0f, statistics +%.0f' % (q((t7 - t2).clamp(min=0), .5) if (t7 > 0).all() else -1, q(t6 - torch.maximum(t7, t2), .5), q(t3 - t6, .5))
    print('%-30s wgs %5d  event %6.1f us  span %6.1f us  %5.1f TF  clk %4.0f MHz | cycles med(p90): prologue %5.0f(%5.0f) main %6.0f(%6.0f) '
          'epilogue %5.0f(%5.0f) | us: %.1f + %.1f + %.1f = %.1f per wg | starts: p50 %.1f p90 %.1f us, %d wgs started after the first ended'
          % (tag, n, ev_us, span, flops / span / 1e6, clk, q(pro, .5), q(pro, .9), q(main, .5), q(main, .9), q(epi, .5), q(epi, .9),
             us(q(pro, .5)), us(q(main, .5)), us(q(epi, .5)), us(q(pro, .5) + q(main, .5) + q(epi, .5)),
             q(starts, .5), q(starts, .9), late) + extra, flush=True)


for (M, I, O) in SHAPES:
    x = torch.randn(M, I, device=dev); w = torch.randn(O, I, device=dev) / math.sqrt(I)
    ldo = (O + 3) // 4 * 4
    y = torch.randn(M, ldo, device=dev); g = torch.randn(M, ldo, device=dev); gin = torch.empty(M, I, device=dev)
    sc, sh = vec(I), vec(I)
    p, q_, mu, r = vec(O), vec(O), vec(O), vec(O)
    a2, b2, mn, iv = vec(I), vec(I), vec(I), vec(I)
    flops = 2.0 * M * I * O
    if 'fwd' in which:
        nb = l.pq_conv1x1_blocks_k(M, O, I, 0); ldp = (O + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        a = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), I, 2), w.data_ptr(), None, y.data_ptr(), part.data_ptr(),
                           M, I, O, ldo, ldp, 1, 0, 0, 0, 0)
        pl = abi.GemmPlan(); abi.check(l.pq_conv1x1_fwd_plan(C.byref(a), C.byref(pl)))
        if pl.family == 2:
            report('fwd   M=%d I=%d O=%d %dx%d S%d' % (M, I, O, pl.bm, pl.bn, pl.stages), lambda: abi.check(l.pq_conv1x1_fwd(C.byref(a), st)), flops)
    if 'dgrad' in which:
        nb = l.pq_conv1x1_blocks_k(M, I, O, 1); ldp = (I + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        wt = torch.zeros(I, ldo, device=dev); wt[:, :O] = w.t()
        a = abi.Conv1x1Dgrad(abi.GSrc(g.data_ptr(), y.data_ptr(), p.data_ptr(), q_.data_ptr(), mu.data_ptr(), r.data_ptr(), ldo, ldo), w.data_ptr(),
                             abi.GSink(gin.data_ptr(), x.data_ptr(), a2.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), I, I, 2, 0, 1, ldp),
                             M, I, O)
        a.wt = wt.data_ptr()
        pl = abi.GemmPlan(); abi.check(l.pq_conv1x1_dgrad_plan(C.byref(a), C.byref(pl)))
        if pl.family == 2:
            report('dgrad M=%d I=%d O=%d %dx%d S%d' % (M, I, O, pl.bm, pl.bn, pl.stages), lambda: abi.check(l.pq_conv1x1_dgrad(C.byref(a), st)), flops)
