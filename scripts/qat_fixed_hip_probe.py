"""GPU probe: HIP QAT forward vs the QAT oracle, both loaded with the G9b state (fixed qparams, observers off): fraction of
values that moved and by how many quantisation steps, for EVERY observed layer (finds the first layer that diverges)."""
import io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
os.environ.setdefault('PQ_QAT_FUSE', '0')
import numpy as np, torch
from oracle import qat
from pqdet_amd import tools
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
from seedfill import seeded_fill, seeded_images
from test_golden_r2 import _targets

g = np.load(os.path.join(ROOT, 'tests/golden/golden_r2.npz'))
names = json.load(open(os.path.join(ROOT, 'tests/golden/g_r2_names.json')))
obs = dict(zip(names['g9b_observers'], g['g9b_observers']))
flat = torch.from_numpy(g['g9b_bn_stats'])
cfg = zoo.cfg_text('mobilenetv2-fpn')
# ---- oracle
net = qat.OracleQATNet(io.StringIO(cfg)); seeded_fill(net.float_net, 0); net.train()
sd = net.float_net.state_dict(); off = 0
with torch.no_grad():
    for k in names['g9b_bn_keys']:
        t = sd[k.replace('.conv.bn.', '.bn.')]; t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
def setfq(fq, row):
    fq.min_val.fill_(row[0]); fq.max_val.fill_(row[1]); fq.scale.fill_(row[2]); fq.zero_point.fill_(int(row[3])); fq.observer_enabled = False
setfq(net.input_fq, obs['qstub.activation_post_process'])
taps = {}
for i, m in enumerate(net.module_list):
    if isinstance(net.w_fq[i], qat.FakeQuant): setfq(net.w_fq[i], obs['module_list.%d.conv.weight_fake_quant' % i])
    if isinstance(net.a_fq[i], qat.FakeQuant):
        kind = 'conv' if m.kind == 'convolutional' else 'ffunc'
        setfq(net.a_fq[i], obs['module_list.%d.%s.activation_post_process' % (i, kind)])
        net.a_fq[i].register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out.detach()))
x, tgt = seeded_images(2, 256, 256, 2), _targets(2, 256, 20, 3)
want = net(x, tgt)
# ---- HIP
hip = DetectionModel(io.StringIO(cfg), quant=True); seeded_fill(hip, 0); hip = hip.cuda(); hip.train()
tools.prepare_qat(tools.fuse_model(hip), backend='qnnpack')
own = hip.plain_state_dict(); off = 0
with torch.no_grad():
    for k in names['g9b_bn_keys']:
        t = own[k.replace('.conv.bn.', '.bn.')]; t.copy_(flat[off:off + t.numel()].view_as(t)); off += t.numel()
    for key, row in hip._qat_index.items():
        if key[0] == 'in': name = 'qstub.activation_post_process'
        elif key[0] == 'w': name = 'module_list.%d.conv.weight_fake_quant' % key[1]
        else: name = 'module_list.%d.%s.activation_post_process' % (key[1], 'conv' if hip.module_list[key[1]]._type == 'convolutional' else 'ffunc')
        hip._qat_state[row] = torch.tensor(obs[name], dtype=torch.float32)
tools.disable_observer(hip); hip._plans.clear()
out = hip(x.cuda(), tuple(t.cuda() for t in tgt)); torch.cuda.synchronize()
print('loss hip', [out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')])
print('loss ora', [want[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')])
plan = [p for p in hip._plans.values() if p.with_loss][-1]
for i in sorted(taps):
    t = plan.tensors[i]
    got = t.raw[:, :t.C].reshape(-1).cpu()
    w = taps[i].permute(0, 2, 3, 1).reshape(-1)
    kind = 'conv' if hip.module_list[i]._type == 'convolutional' else 'ffunc'
    step = obs['module_list.%d.%s.activation_post_process' % (i, kind)][2]
    d = (got - w).abs()
    print('layer %3d %-13s C=%4d moved %.5f max steps %.1f' % (i, hip.module_list[i]._type, t.C, (d > 0.25 * step).float().mean().item(), (d.max() / step).item()))
