"""debug: pq_conv1x1_dgrad without transposed weights at (65536, I=256, O=75)"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, I, O in ((65536, 256, 75), (65536, 256, 76), (65536, 256, 64), (4096, 256, 75), (65536, 128, 75)):
    g = torch.Generator().manual_seed(1)
    ldg = (O + 3) // 4 * 4
    gy = torch.zeros(M, ldg); gy[:, :O] = torch.randn(M, O, generator=g)
    w = torch.randn(O, I, generator=g) / math.sqrt(O)
    want = gy[:, :O] @ w
    gyd, wd = gy.cuda(), w.cuda()
    out = torch.full((M, I), float('nan'), device='cuda')
    a = abi.Conv1x1Dgrad()
    a.dy = abi.GSrc(gyd.data_ptr(), None, None, None, None, None, ldg, ldg)
    a.w = wd.data_ptr()
    a.dx = abi.GSink(out.data_ptr(), None, None, None, None, None, None, I, I, 0, 0, 1, 0)
    a.M, a.I, a.O = M, I, O
    pl = abi.GemmPlan(); abi.check(l.pq_conv1x1_dgrad_plan(C.byref(a), C.byref(pl)))
    abi.check(l.pq_conv1x1_dgrad(C.byref(a), st)); torch.cuda.synchronize()
    err = (out.cpu() - want).abs()
    bad = err > 1e-3
    print((M, I, O), pl.key(), 'max err %.3g, bad %d' % (err.max().item(), int(bad.sum())))
    if bad.any():
        rows = bad.any(1).nonzero()[:, 0]; cols = bad.any(0).nonzero()[:, 0]
        print('  bad rows: %d (first %s, mod128 %s)' % (len(rows), rows[:8].tolist(), sorted(set((rows % 128).tolist()))[:20]))
        print('  bad cols: %d (first %s)' % (len(cols), cols[:16].tolist()))
