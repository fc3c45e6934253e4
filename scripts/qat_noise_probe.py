"""Noise band of the QAT fixtures: the CPU oracle on THIS machine vs the G9 fixtures (recorded on another CPU)."""
from _dbg_common import *
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import qat as oqat
from test_oracle_qat import _g9, qat_batch, qat_observers
g, names = _g9()
net = oqat.OracleQATNet(io.StringIO(zoo.cfg_text('mobilenetv2-fpn'))); seeded_fill(net.float_net, 0); net.train()
for step in range(2):
    x, tgt = qat_batch(step); net.zero_grad(); out = net(x, tgt); out['loss'].mean().backward()
    print('step', step, [round(out[k].item(), 3) for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')], g['g9_loss'][step][:4])
obs = qat_observers(net)
if names['observers'][-1].startswith('qstub'): obs = np.concatenate([obs[1:], obs[:1]])
want = g['g9_observers']
rng = np.maximum(np.abs(want[:, 0]), np.abs(want[:, 1]))
dev = np.abs(obs[:, :2] - want[:, :2]).max(1) / (rng + 1e-6)
idx = np.argsort(-dev)[:8]
for i in idx: print(names['observers'][i], obs[i, :2], want[i, :2], 'dev %.3f' % dev[i])
print('rows with dev > 5%%: %d, > 15%%: %d of %d' % ((dev > 0.05).sum(), (dev > 0.15).sum(), len(dev)))
