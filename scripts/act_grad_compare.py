"""Debug helper: gradient w.r.t. layer outputs (where the layer has no activation), HIP engine vs CPU oracle."""
from _dbg_common import *
import torch.nn.functional as F
from oracle import loss as oloss

B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
text = zoo.cfg_text('mobilenetv2-fpn')
net = DetectionModel(io.StringIO(text)); seeded_fill(net, 3)
ref = OracleNet(io.StringIO(text)); ref.load_state_dict(net.state_dict())
net = net.cuda(); net.train(); ref.train()
x = seeded_images(B, H, W, 5); tgt = targets(B, H, W, 20, 7)
out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
plan = list(net._plans.values())[0]

# oracle forward replicated with hooks
store, fwd = {}, {}
cache, heads = [], []
xx = x
for i, m in enumerate(ref.module_list):
    kind = m.kind
    if kind == 'convolutional':
        xx = m(xx)
    elif kind == 'shortcut':
        xx = xx + cache[m.spec['from']]
    elif kind == 'route':
        xx = cache[m.idx[0]] if not isinstance(m.spec['layers'], list) else torch.cat([cache[j] for j in m.idx], 1)
    elif kind == 'upsample':
        xx = F.interpolate(xx, scale_factor=2, mode='nearest')
    elif kind == 'yolo':
        opt = m.opt
        pick = {8: (0, 3), 16: (1, 4), 32: (2, 5)}[opt['stride']]
        heads.append(oloss.loss_per_scale(ref.decode(xx, opt['classes'], opt['stride']), tgt[pick[0]], tgt[pick[1]], opt))
    if kind != 'yolo' and xx.requires_grad:
        xx.register_hook(lambda g, i=i: store.__setitem__(i, g))
        fwd[i] = xx
    cache.append(xx)
total = sum(h[0] for h in heads)
total.mean().backward()
for i, m in enumerate(net.module_list):
    t = plan.tensors[i]
    if m._type == 'yolo' or t.grad is None or i not in store:
        continue
    if t.act != 0:
        continue
    if m._type == 'route' and m.single:
        continue
    g = t.grad.view(t.B, t.H, t.W, t.ld)[..., :t.C].cpu().double()
    w = store[i].permute(0, 2, 3, 1).double()
    e = ((g - w).norm() / (w.norm() + 1e-30)).item()
    # forward value check (raw for identity tensors; bn tensors compare post-affine)
    raw = t.raw.view(t.B, t.H, t.W, t.ld)[..., :t.C]
    if t.bn is not None:
        raw = raw * t.bn.scale + t.bn.shift
    fe = ((raw.cpu().double() - fwd[i].detach().permute(0, 2, 3, 1).double()).norm() / fwd[i].norm().item()).item()
    print('%3d %-14s C=%4d %3dx%-3d consumers %d  fwd relerr %.2e   grad relerr %.3e' % (i, m._type, t.C, t.H, t.W, t.n_consumers, fe, e))
