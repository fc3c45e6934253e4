"""One shape of the walking depthwise backward (pq_dw3x3_bwd part 5), a few launches: for rocprofv3 / counter runs.
    python scripts/dw_walk_one.py B H W C stride [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
B, H, W, Cc, s = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); dev = 'cuda'
OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
x = torch.randn(B, H, W, Cc, device=dev); yo = torch.randn(B, OH, OW, Cc, device=dev); g = torch.randn(B, OH, OW, Cc, device=dev)
w = torch.randn(Cc, 9, device=dev); gin = torch.empty(B, H, W, Cc, device=dev)
a, b2, mn, iv, p, q, mu, r = (torch.rand(Cc + 32, device=dev) + 0.5 for _ in range(8))
nbw = l.pq_dw3x3_bwd_walk_blocks(B, H, W, Cc, s)
partw = torch.empty(nbw * 2 * Cc, device=dev); scw = torch.empty(nbw * Cc * 9 + 16, device=dev)
gs = abi.GSrc(g.data_ptr(), yo.data_ptr(), p.data_ptr(), q.data_ptr(), mu.data_ptr(), r.data_ptr(), Cc, Cc)
src = abi.Src(x.data_ptr(), a.data_ptr(), b2.data_ptr(), Cc, 2)
sinkw = abi.GSink(gin.data_ptr(), x.data_ptr(), a.data_ptr(), b2.data_ptr(), mn.data_ptr(), iv.data_ptr(), partw.data_ptr(), Cc, Cc, 2, 0, 1, Cc)
f5 = abi.DwBwd(gs, src, w.data_ptr(), sinkw, None, scw.data_ptr(), B, H, W, OH, OW, Cc, s, 5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
abi.check(l.pq_dw3x3_bwd(C.byref(f5), st)); torch.cuda.synchronize(); e0.record()
for _ in range(reps): abi.check(l.pq_dw3x3_bwd(C.byref(f5), st))
e1.record(); torch.cuda.synchronize()
print('%d rows, %.3f ms per launch' % (nbw, e0.elapsed_time(e1) / reps))
