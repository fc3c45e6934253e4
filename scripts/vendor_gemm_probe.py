"""Reference point: what the vendor library (rocBLAS / hipBLASLt through torch.mm, fp32) reaches on the 1x1-conv GEMM shapes of
mobilenetv2-fpn -- plain C = A @ B^T without any of the fused prologues / epilogues.  Not used by the product."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(1048576, 16, 96), (262144, 24, 144), (65536, 32, 192), (65536, 128, 256), (16384, 64, 384), (16384, 96, 576), (16384, 256, 512),
          (16384, 512, 256), (4096, 160, 960), (4096, 320, 1280), (4096, 1280, 512), (4096, 512, 1024), (4096, 1024, 512)]
for M, K, N in shapes:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(N, K, device='cuda')
    for _ in range(5): c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    # wgrad shape: [N x M] @ [M x K]
    g = torch.randn(M, N, device='cuda')
    for _ in range(5): w = g.t() @ a
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): w = g.t() @ a
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 30
    fl = 2.0 * M * K * N
    print('M=%8d K=%4d N=%4d | A@B^T %.3f ms %5.1f TF/s | G^T@A (wgrad shape) %.3f ms %5.1f TF/s' % (M, K, N, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9), flush=True)
