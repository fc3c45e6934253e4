"""Debug helper: pre-activation value / gradient of one conv layer, HIP engine vs the fp64 and fp32 CPU oracles."""
from _dbg_common import *
import copy
name, L, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
text = zoo.cfg_text(name)
net = DetectionModel(io.StringIO(text)); seeded_fill(net, 3)
ref = OracleNet(io.StringIO(text)); ref.load_state_dict(net.state_dict())
ref64 = copy.deepcopy(ref).double()
net = net.cuda(); net.train(); ref.train(); ref64.train()
x = seeded_images(B, H, W, 5); tgt = targets(B, H, W, 20, 7)
out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
plan = list(net._plans.values())[0]
cap = {}
def hook(tag):
    def f(mod, inp, outp):
        cap[tag + '_u'] = outp.detach()
        outp.register_hook(lambda g: cap.__setitem__(tag + '_g', g.detach()))
    return f
ref.module_list[L].bn.register_forward_hook(hook('r32'))
ref64.module_list[L].bn.register_forward_hook(hook('r64'))
ref(x, tgt)['loss'].mean().backward()
ref64(x.double(), tuple(t.double() for t in tgt))['loss'].mean().backward()
t = plan.tensors[L]
raw = t.raw.view(t.B, t.H, t.W, t.ld)[..., :t.C]
u = (raw * t.bn.scale + t.bn.shift).cpu().double()
g = t.grad.view(t.B, t.H, t.W, t.ld)[..., :t.C].cpu().double()
u64, g64 = cap['r64_u'].permute(0, 2, 3, 1), cap['r64_g'].permute(0, 2, 3, 1)
u32, g32 = cap['r32_u'].permute(0, 2, 3, 1).double(), cap['r32_g'].permute(0, 2, 3, 1).double()
print('u   : hip-vs-64 %.3e   ref32-vs-64 %.3e' % ((u - u64).norm() / u64.norm(), (u32 - u64).norm() / u64.norm()))
print('g   : hip-vs-64 %.3e   ref32-vs-64 %.3e' % ((g - g64).norm() / g64.norm(), (g32 - g64).norm() / g64.norm()))
d = (g - g64).abs()
thr = 1e-3 * g64.abs().max()
bad = (d > thr).nonzero()
print('elements off by > 1e-3*max: %d of %d (hip), %d (ref32)' % (len(bad), g.numel(), int(((g32 - g64).abs() > thr).sum())))
for idx in bad[:20]:
    i = tuple(idx.tolist())
    print(i, 'u hip %.6e u64 %.6e | g hip %.4e g64 %.4e g32 %.4e' % (u[i], u64[i], g[i], g64[i], g32[i]))
s1h, s1r, s1 = g.sum((0, 1, 2)), g32.sum((0, 1, 2)), g64.sum((0, 1, 2))
print('s1 err hip %.3e ref32 %.3e' % ((s1h - s1).norm() / s1.norm(), (s1r - s1).norm() / s1.norm()))
ch = (s1h - s1).abs().argsort(descending=True)[:8]
for c in ch.tolist():
    print('ch %d s1 hip %.6e ref64 %.6e ref32 %.6e  sum|g| %.3e' % (c, s1h[c], s1[c], s1r[c], g64[..., c].abs().sum()))
