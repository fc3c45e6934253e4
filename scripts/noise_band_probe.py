"""Which parameters of a small train step leave the fp32 noise band of tests/test_model_gpu.py, per engine variant.
usage: python scripts/noise_band_probe.py [name B H W]   (env switches are toggled in-process, one plan per variant)"""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
import torch
from test_model_gpu import _pair, _targets, _relerr, seeded_images, DEV

name = sys.argv[1] if len(sys.argv) > 1 else 'regnetx-600m-fpn'
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 128, 160)
x = seeded_images(B, H, W, 5)
tgt = _targets(B, H, W, 20, seed=7)
_, ref = _pair(seed=3, name=name)
ref64 = copy.deepcopy(ref).double()
ref.train(), ref64.train()
ref(x, tgt)['loss'].mean().backward()
ref64(x.double(), tuple(t.double() for t in tgt))['loss'].mean().backward()
for env in ({}, {'PQ_BN_ATOMIC': '0'}):
    for k in ('PQ_BN_ATOMIC', 'PQ_FWD_SIDE'):
        os.environ.pop(k, None)
    os.environ.update(env)
    net, _ = _pair(seed=3, name=name)
    net.train()
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    torch.cuda.synchronize()
    rows, eh, er = [], [], []
    for (n1, p1), (_, p2), (_, p3) in zip(net.named_parameters(), ref.named_parameters(), ref64.named_parameters()):
        if p3.grad.norm().item() < 1e-6:
            continue
        e_hip, e_ref = _relerr(p1.grad, p3.grad), _relerr(p2.grad, p3.grad)
        eh.append(e_hip); er.append(e_ref)
        if e_hip > max(5 * e_ref, 2e-3):
            rows.append('%s hip %.2e ref %.2e' % (n1, e_hip, e_ref))
    print(env or 'default', 'loss %.6f' % out['loss'].item(), '| outside 5x band:', len(rows))
    eh, er = sorted(eh), sorted(er)
    print('    gradient error vs fp64 over %d parameters: hip median %.2e max %.2e | reference fp32 median %.2e max %.2e'
          % (len(eh), eh[len(eh) // 2], eh[-1], er[len(er) // 2], er[-1]))
    for r in rows:
        print('    ', r)
