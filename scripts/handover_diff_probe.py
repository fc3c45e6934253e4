"""HIP vs HIP: the statistics hand-over (PQ_BN_ATOMIC=1, default) against the finalize launches (=0) on one small train step --
stored BatchNorm vectors, running statistics and every parameter gradient, layer by layer.
usage: python scripts/handover_diff_probe.py [name B H W [ENVVAR=a,b]]   (default PQ_BN_ATOMIC=1,0; e.g. PQ_BN_BWD_ATOMIC=1,0)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
import torch
from test_model_gpu import _pair, _targets, _relerr, seeded_images, DEV

name = sys.argv[1] if len(sys.argv) > 1 else 'regnetx-600m-fpn'
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 128, 160)
x = seeded_images(B, H, W, 5)
tgt = _targets(B, H, W, 20, seed=7)
var, modes = (sys.argv[5].split('=')[0], sys.argv[5].split('=')[1].split(',')) if len(sys.argv) > 5 else ('PQ_BN_ATOMIC', ['1', '0'])
res = {}
for mode in modes:
    os.environ[var] = mode
    net, _ = _pair(seed=3, name=name)
    net.train()
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    torch.cuda.synchronize()
    plan = list(net._plans.values())[0]
    vec = {}
    for i, t in enumerate(plan.tensors):
        if t is not None and getattr(t, 'bn', None) is not None:
            b = t.bn
            for k in ('mean', 'invstd', 'scale', 'shift', 'p', 'q', 'r'):
                v = getattr(b, k, None)
                if v is not None:
                    vec['%d.%s' % (i, k)] = v[:b.C].detach().cpu().double().clone()
    res['1' if mode == modes[0] else '0'] = (vec, {n: p.grad.detach().cpu().double().clone() for n, p in net.named_parameters()},
                 {n: b.detach().cpu().double().clone() for n, b in net.named_buffers() if b.is_floating_point()},
                 out['loss'].item(), (getattr(plan, 'bn_handover', None), getattr(plan, 'bn_bwd_handover', None)))
print('loss', res['1'][3], res['0'][3], 'hand-overs', res['1'][4], res['0'][4])
for title, idx, thr in (('stored vectors', 0, 1e-5), ('gradients', 1, 1e-3), ('buffers', 2, 1e-5)):
    a, b = res['1'][idx], res['0'][idx]
    bad = [(k, ((a[k] - b[k]).norm() / (b[k].norm() + 1e-30)).item(), (a[k] - b[k]).abs().max().item()) for k in b]
    bad = [r for r in bad if r[1] > thr]
    print(title, ': %d of %d differ by more than %g (relative norm)' % (len(bad), len(b), thr))
    for r in bad[:60]:
        print('    %-40s rel %.3e  max abs %.3e' % r)
