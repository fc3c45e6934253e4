"""Per-parameter gradient error of the HIP step and of the fp32 oracle, both against the fp64 oracle."""
from _dbg_common import *
import copy
name, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
text = zoo.cfg_text(name)
net = DetectionModel(io.StringIO(text)); seeded_fill(net, 3)
ref = OracleNet(io.StringIO(text)); ref.load_state_dict(net.state_dict())
ref64 = copy.deepcopy(ref).double()
net = net.cuda(); net.train(); ref.train(); ref64.train()
if len(sys.argv) > 5:
    net.overlap_wgrad = bool(int(sys.argv[5]))
x = seeded_images(B, H, W, 5); tgt = targets(B, H, W, 20, 7)
out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
want = ref(x, tgt); want['loss'].mean().backward()
w64 = ref64(x.double(), tuple(t.double() for t in tgt)); w64['loss'].mean().backward()
print('loss', out['loss'].item(), want['loss'].item(), w64['loss'].item())
rows = []
for (n1, p1), (_, p2), (_, p3) in zip(net.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
    c = p3.grad.double()
    nb = c.norm().item()
    eh = ((p1.grad.cpu().double() - c).norm() / (nb + 1e-30)).item()
    er = ((p2.grad.double() - c).norm() / (nb + 1e-30)).item()
    rows.append((eh / max(er, 1e-12), eh, er, nb, n1))
rows.sort(reverse=True)
for r in rows[:25]:
    print('ratio %8.1f  hip %.2e  ref32 %.2e  |g| %.2e  %s' % r)
print('max hip', max(r[1] for r in rows), 'max ref', max(r[2] for r in rows))
