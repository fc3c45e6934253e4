"""One GPU: wall time of the backward replayed in one piece against the bucketed replay the data-parallel path uses (8 MiB gradient
buckets handed to a communication stream at each mark).  The reducer here does everything but the collective itself (the
communication stream's waits on both compute streams are real), so the difference is what the bucketed hand-off costs the
compute streams.  usage: python scripts/ddp_bucket_wall.py [batch size]"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
from pqdet_amd.parallel import GradReducer

class LocalReducer(GradReducer):
    def __init__(self, bucket_bytes=8 << 20):
        self.bucket_elems, self.group, self.world, self.stream, self.handles = max(1, bucket_bytes // 4), None, 2, None, []
        self.calls = 0
    def reduce_range(self, flat_grad, lo, hi, also_wait=None):
        if hi <= lo: return
        if self.stream is None: self.stream = torch.cuda.Stream()
        ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream()); self.stream.wait_event(ready)
        if also_wait is not None: self.stream.wait_stream(also_wait)
        with torch.cuda.stream(self.stream): flat_grad[lo:hi].mul_(1.0)        # stands in for the all-reduce (touches the range)
        self.calls += 1

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn'))).cuda(); net.train()
x = torch.randn(B, 3, 512, 512, device='cuda'); tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, 512, 20, seed=1))
def step():
    net.zero_grad(); net(x, tgt)['loss'].mean().backward()
def wall(n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
plain = wall()
red = LocalReducer(); net._reducer = red
bucketed = wall()
print('forward + backward, plain: %.3f ms; bucketed (%d hand-offs per step): %.3f ms; difference %+.2f %%' % (
    plain, red.calls // 35, bucketed, (bucketed - plain) / plain * 100))
