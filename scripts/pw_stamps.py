"""Where the fused narrow-layer backward (pq_conv1x1_bwd) spends its time: per-workgroup phase cycles of the DIAGNOSTIC build
(python -m pqdet_amd.build -DPQ_STAMPS --tag=stamps; PQDET_HIP_LIB=pqdet_amd/lib/libpqdet_hip_stamps.so python scripts/pw_stamps.py).
Per shape: workgroups, tiles per workgroup, in-kernel clock, and per TILE the median cycles a workgroup waits at the tile's first
barrier, stages the operands (global -> registers -> LDS), contracts (both gradients) and runs the sink (LDS -> global)."""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi

SHAPES = [(1048576, 16, 96), (1048576, 32, 16), (262144, 24, 144), (262144, 144, 24), (262144, 96, 24)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
l = abi.lib()
l.pq_debug_set_stamps_pw.restype = C.c_int
l.pq_debug_set_stamps_pw.argtypes = [C.c_void_p]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = 'cuda'
NWG = 4096
stamps = torch.zeros(NWG * 8, dtype=torch.int64, device=dev)
assert l.pq_debug_set_stamps_pw(stamps.data_ptr()) == 0


def vec(n):
    return torch.rand(n + 32, device=dev) + 0.5


for (M, I, O) in SHAPES:
    nb = l.pq_conv1x1_bwd_blocks(M, I, O)
    if nb <= 0:
        continue
    for accumulate in (0, 1):
        x = torch.randn(M, I, device=dev); w = torch.randn(O, I, device=dev) / math.sqrt(I)
        y = torch.randn(M, O, device=dev); g = torch.randn(M, O, device=dev); gin = torch.zeros(M, I, device=dev)
        p, q_, mu, r = vec(O), vec(O), vec(O), vec(O)
        a2, b2, mn, iv = vec(I), vec(I), vec(I), vec(I)
        ldp = (I + 15) // 16 * 16
        part = torch.empty(nb * 2 * ldp, device=dev)
        scratch = torch.empty(l.pq_conv1x1_bwd_scratch_bytes(M, I, O) // 4 + 16, device=dev)
        dw = torch.empty(O, I, device=dev)
        a = abi.Conv1x1Bwd(abi.GSrc(g.data_ptr(), y.data_ptr(), p.data_ptr(), q_.data_ptr(), mu.data_ptr(), r.data_ptr(), O, O),
                           w.data_ptr(), abi.Src(x.data_ptr(), a2.data_ptr(), b2.data_ptr(), I, 2),
                           abi.GSink(gin.data_ptr(), x.data_ptr(), a2.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), part.data_ptr(), I, I, 2,
                                     accumulate, 1, ldp),
                           dw.data_ptr(), scratch.data_ptr(), M, I, O)
        fn = lambda: abi.check(l.pq_conv1x1_bwd(C.byref(a), st))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        stamps.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ev_us = e0.elapsed_time(e1) * 1e3
        s = stamps.view(NWG, 8).cpu()
        s = s[s[:, 6] != 0].double()
        n = s.shape[0]
        nt = s[:, 6]
        span = (s[:, 5].max() - s[:, 4].min()).item() / 100.0
        clk = (s[:, 7] / (s[:, 5] - s[:, 4]).clamp(min=1)).median().item() * 100.0
        med = lambda v: v.median().item()
        tb = 4.0 * M * (2 * O + (3 + accumulate) * I)
        print('M=%d I=%d O=%d acc=%d  wgs %d x %.0f tiles  event %.1f us  span %.1f us  %.0f GB/s  clk %.0f MHz | per tile, median cycles: '
              'barrier wait %.0f  stage %.0f  contract %.0f  sink %.0f  (sum %.1f us; wg total %.1f us, outside the tile loop %.1f us)'
              % (M, I, O, accumulate, n, med(nt), ev_us, span, tb / span / 1e3, clk, med(s[:, 0] / nt), med(s[:, 1] / nt), med(s[:, 2] / nt),
                 med(s[:, 3] / nt), med((s[:, 0] + s[:, 1] + s[:, 2] + s[:, 3]) / nt) / clk, med(s[:, 7]) / clk,
                 med(s[:, 7] - s[:, 0] - s[:, 1] - s[:, 2] - s[:, 3]) / clk), flush=True)
