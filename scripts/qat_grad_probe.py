"""Debug helper: per-parameter gradient error of the QAT step (fake-quant on/off) vs the CPU oracle."""
from _dbg_common import *
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import qat as oqat
from pqdet_amd import tools
from test_oracle_qat import qat_batch
fq_on = int(sys.argv[1]) if len(sys.argv) > 1 else 0
text = zoo.cfg_text('mobilenetv2-fpn')
net = DetectionModel(io.StringIO(text), quant=True); seeded_fill(net, 4); net = net.cuda(); net.train(); tools.prepare_qat(net)
net.qat_fake_quant_enabled = bool(fq_on)
ref = oqat.OracleQATNet(io.StringIO(text)); seeded_fill(ref.float_net, 4); ref.train()
for m in ref.modules():
    if isinstance(m, oqat.FakeQuant): m.fake_quant_enabled = bool(fq_on)
x, tgt = qat_batch(0, B=3, S=160)
out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
want = ref(x, tgt); want['loss'].mean().backward()
print('loss', out['loss'].item(), want['loss'].item())
mine = dict(net.named_parameters())
for k, p in ref.float_net.named_parameters():
    gw = p.grad.double(); gm = mine[k].grad.cpu().double()
    e = ((gm - gw).norm() / (gw.norm() + 1e-30)).item()
    if e > 0.05 or k.startswith('module_list.0.') or k.startswith('module_list.2.') or k.startswith('module_list.101'):
        print('%-34s err %.3e  |ref| %.3e |hip| %.3e  ratio %.4f' % (k, e, gw.norm().item(), gm.norm().item(), gm.norm().item() / (gw.norm().item() + 1e-30)))
