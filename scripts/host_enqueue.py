"""Host-side cost of enqueueing one train step (GPU idle at the start, no sync inside): if this approaches the GPU step time
the step is launch-bound somewhere.  usage: host_enqueue.py [batch size]"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import optim
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
model = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('MODEL', 'mobilenetv2-fpn')))).cuda(); model.train()
model.nan_check = 'deferred'
opt = optim.Adam(model.parameters(), lr=4e-4, weight_decay=0.0)
x = torch.randn(B, 3, S, S).cuda()
tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, S, 20, seed=1))

def step(marks=None):
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    out = model(x, tgt); t.append(time.perf_counter())
    loss = out['loss'].mean(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if marks is not None: marks.append([1e3 * (b - a) for a, b in zip(t, t[1:])])

for _ in range(10): step()
torch.cuda.synchronize()
marks = []
for _ in range(8):
    torch.cuda.synchronize()
    step(marks)
torch.cuda.synchronize()
print('host ms per phase (GPU idle at start): zero_grad, forward, loss.mean, backward, opt.step | total')
for m in marks: print(' '.join('%7.3f' % v for v in m), '| %7.3f' % sum(m))
t0 = time.perf_counter()
for _ in range(40): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print('40 steps back to back: host loop returns after %.2f ms/step, GPU done after %.2f ms/step' % (th / 40 * 1e3, tg / 40 * 1e3))
