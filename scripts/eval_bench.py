"""Inference pipeline timing: forward, recover, NMS for a batch of 16 at 512x512 (HIP events).
The score threshold is chosen so that a random-init model passes a trained-model-like number of candidates."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import ops, tools
from pqdet_amd.dataset import RECOVER_BBOXES_REGISTER
from pqdet_amd.eval import Evaluator
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
torch.manual_seed(0)
net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn'))).cuda(); net.eval()
x = torch.randn(16, 3, 512, 512).cuda()
shapes = torch.tensor([[375., 500.]] * 16)
with torch.no_grad(): pred = net(x)
score = (pred[..., 5:] * pred[..., 4:5]).flatten()
for target in (300, 3000, 30000):
    thr = float(score.kthvalue(score.numel() - target * 16).values)
    ev = Evaluator(net, dataset_name='voc', classes=['c%d' % i for i in range(20)], input_size=(512, 512), score_threshold=thr, iou_threshold=0.45)
    for _ in range(3): dets = ev.detect(x, shapes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dets = ev.detect(x, shapes)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        with torch.no_grad(): p = net(x)
    e1.record(); torch.cuda.synchronize()
    fwd = e0.elapsed_time(e1) / 10
    print('~%d candidates/image: detect() %.2f ms per batch of 16 (forward %.2f ms) -> %.0f img/s, kept %s' % (
        target, dt * 1e3, fwd, 16 / dt, [len(d) for d in dets][:4]), flush=True)
