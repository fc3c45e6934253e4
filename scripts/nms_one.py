import sys, torch
sys.path.insert(0, '/root/repo')
from pqdet_amd import ops
g = torch.Generator().manual_seed(1)
N, C, want = 16128, 20, 1000
xy = torch.rand(N, 2, generator=g) * 480; wh = torch.rand(N, 2, generator=g) * 120 + 8
sc = torch.rand(N, C, generator=g) * 0.0999
hot = torch.randperm(N * C, generator=g)[:want]; sc.view(-1)[hot] = 0.1001 + 0.8999 * torch.rand(want, generator=g)
d = torch.cat([xy, xy + wh, sc], 1).contiguous().cuda()
for _ in range(20): dets, cnt = ops.nms(d, 0.1, 0.45)
torch.cuda.synchronize()
