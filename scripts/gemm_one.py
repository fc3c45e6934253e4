"""Run pq_conv1x1_fwd a few times on given shapes (for rocprofv3 --pmc).  usage: gemm_one.py M I O [M I O ...]"""
import ctypes as C, os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
args = list(map(int, sys.argv[1:]))
for k in range(0, len(args), 3):
    M, I, O = args[k:k + 3]
    x = torch.randn(M, I, device='cuda'); w = torch.randn(O, I, device='cuda') / math.sqrt(I)
    y = torch.empty(M, O, device='cuda'); sc = torch.rand(I + 32, device='cuda') + .5; sh = torch.rand(I + 32, device='cuda')
    nb = l.pq_conv1x1_blocks_k(M, O, I, 0); ldp = (O + 15) // 16 * 16
    part = torch.empty(nb * 2 * ldp, device='cuda')
    a = abi.Conv1x1Fwd(abi.Src(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), I, 2), w.data_ptr(), None, y.data_ptr(), part.data_ptr(), M, I, O, O, ldp, 1, 0, 0, 0, 0)
    for _ in range(5):
        abi.check(l.pq_conv1x1_fwd(C.byref(a), st))
    torch.cuda.synchronize()
