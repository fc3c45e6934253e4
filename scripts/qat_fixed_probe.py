"""CPU probe: the QAT ORACLE loaded with the G9b state (fixed qparams, observers off) against the reference's recorded
activations -- how far do two correct fp32 evaluations drift apart through a fake-quantised train-mode network?"""
import io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from oracle import qat
from pqdet_amd.model import zoo
from seedfill import seeded_fill, seeded_images
from test_golden_r2 import _targets

g = np.load(os.path.join(ROOT, 'tests/golden/golden_r2.npz'))
names = json.load(open(os.path.join(ROOT, 'tests/golden/g_r2_names.json')))
torch.set_num_threads(8)
net = qat.OracleQATNet(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')))
seeded_fill(net.float_net, 0)
net.train()
sd = net.float_net.state_dict()
flat = torch.from_numpy(g['g9b_bn_stats']); off = 0
with torch.no_grad():
    for k in names['g9b_bn_keys']:
        t = sd[k.replace('.conv.bn.', '.bn.')]
        t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
obs = dict(zip(names['g9b_observers'], g['g9b_observers']))
def setfq(fq, row):
    fq.min_val.fill_(row[0]); fq.max_val.fill_(row[1]); fq.scale.fill_(row[2]); fq.zero_point.fill_(int(row[3])); fq.observer_enabled = False
setfq(net.input_fq, obs['qstub.activation_post_process'])
for i, m in enumerate(net.module_list):
    if isinstance(net.w_fq[i], qat.FakeQuant):
        setfq(net.w_fq[i], obs['module_list.%d.conv.weight_fake_quant' % i])
    if isinstance(net.a_fq[i], qat.FakeQuant):
        kind = 'conv' if m.kind == 'convolutional' else 'ffunc'
        setfq(net.a_fq[i], obs['module_list.%d.%s.activation_post_process' % (i, kind)])
taps = {}
orig = {}
for i in (1, 23, 49, 71, 92):
    fq = net.a_fq[i]
    def hook(mod, inp, out, i=i): taps[i] = out.detach()
    fq.register_forward_hook(hook)
out = net(seeded_images(2, 256, 256, 2), _targets(2, 256, 20, 3))
print('loss', [out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')], 'ref', g['g9b_loss'][:4])
for i in (1, 23, 49, 71, 92):
    want = torch.from_numpy(g['g9b_act%d' % i]); stride = int(g['g9b_act%d_stride' % i][0])
    got = taps[i].permute(0, 2, 3, 1).reshape(-1)[::stride]
    step = obs['module_list.%d.conv.activation_post_process' % i][2]
    d = (got - want).abs()
    print(i, 'moved %.4f  max steps %.1f' % ((d > 0.25 * step).float().mean().item(), (d.max() / step).item()))
