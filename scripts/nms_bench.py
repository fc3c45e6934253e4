"""NMS micro-benchmark: one image of (16128, 4+20) recovered predictions with a controlled number of
candidates above the score threshold; device time of pq_nms (HIP events) next to the CPU oracle on the same input,
selection compared bit-exactly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import postprocess as opost
from pqdet_amd import ops

N, C = 16128, 20
# NMS_COUNTS=10000 (comma list) restricts the candidate counts, NMS_NO_CPU=1 skips the CPU oracle (profiling runs)
counts = [int(v) for v in os.environ.get('NMS_COUNTS', '1000,10000,100000').split(',')]
for want in counts:
    g = torch.Generator().manual_seed(want)
    xy = torch.rand(N, 2, generator=g) * 480
    wh = torch.rand(N, 2, generator=g) * 120 + 8
    sc = torch.rand(N, C, generator=g) * 0.0999                      # below the threshold ...
    hot = torch.randperm(N * C, generator=g)[:want]
    sc.view(-1)[hot] = 0.1001 + 0.8999 * torch.rand(want, generator=g)   # ... except `want` candidates
    boxes = torch.cat([xy, xy + wh, sc], 1).contiguous()
    d = boxes.cuda()
    cc = max(8192, 1 << (2 * want - 1).bit_length())           # what the adaptive capacity settles on for this candidate count
    if cc >= N * C: cc = 0
    for _ in range(3):
        dets, cnt = ops.nms(d, 0.1, 0.45, cand_cap=cc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        dets, cnt = ops.nms(d, 0.1, 0.45, cand_cap=cc)
    e1.record(); torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / reps
    k = int(cnt.item())
    if os.environ.get('NMS_NO_CPU'):
        print('candidates %6d -> kept %5d | pq_nms %.3f ms' % (want, k, gpu_ms), flush=True)
        continue
    t0 = time.perf_counter()
    ref = opost.torch_nms(boxes.numpy(), 0.1, 0.45)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = ref.shape == (k, 6) and np.array_equal(dets[:k].cpu().numpy(), ref)
    print('candidates %6d -> kept %5d | pq_nms %.3f ms | CPU oracle %.1f ms | identical selection: %s' % (want, k, gpu_ms, cpu_ms, same), flush=True)

# NMS_BATCH=1: the evaluator's batched call (16 images, capacity 8192 candidates per image: one workgroup per image on the in-house
# LDS sort) at 100 / 1000 / 5000 candidates per image, against the same call forced onto the library radix sorts (cand_cap above it)
if os.environ.get('NMS_BATCH'):
    from pqdet_amd import abi
    B = 16
    for per in (100, 1000, 5000):
        g = torch.Generator().manual_seed(per)
        xy = torch.rand(B, N, 2, generator=g) * 480
        wh = torch.rand(B, N, 2, generator=g) * 120 + 8
        sc = torch.rand(B, N, C, generator=g) * 0.0999
        for b in range(B):
            hot = torch.randperm(N * C, generator=g)[:per]
            sc[b].view(-1)[hot] = 0.1001 + 0.8999 * torch.rand(per, generator=g)
        d = torch.cat([xy, xy + wh, sc], 2).contiguous().cuda()
        res = []
        for cc in (B * 8192, B * 8192 + 4096):          # per-image LDS sort | library sorts
            for _ in range(3):
                dets, img, cnt = ops.nms_batch(d, 0.1, 0.45, cand_cap=cc)
            path = abi.lib().pq_nms_last_path()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dets, img, cnt = ops.nms_batch(d, 0.1, 0.45, cand_cap=cc)
            e1.record(); torch.cuda.synchronize()
            res.append((path, e0.elapsed_time(e1) / 10, int(cnt.item())))
        print('batch of %d images, %5d candidates each -> kept %6d | path %d: %.3f ms | path %d: %.3f ms' % (
            B, per, res[0][2], res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
