"""Debug helper: QAT state (BN running statistics, observers) of the HIP engine vs the CPU oracle after each step."""
from _dbg_common import *
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import qat as oqat
from pqdet_amd import tools
from test_oracle_qat import qat_batch, qat_observers
text = zoo.cfg_text('mobilenetv2-fpn')
net = DetectionModel(io.StringIO(text), quant=True); seeded_fill(net, 0); net = net.cuda(); net.train(); tools.prepare_qat(net)
ref = oqat.OracleQATNet(io.StringIO(text)); seeded_fill(ref.float_net, 0); ref.train()
for step in range(2):
    x, tgt = qat_batch(step)
    out = net(x.cuda(), tuple(t.cuda() for t in tgt)); out['loss'].mean().backward()
    want = ref(x, tgt); ref.zero_grad(); want['loss'].mean().backward()
    print('step', step, 'loss', out['loss'].item(), want['loss'].item())
    sd, rd = net.state_dict(), ref.float_net.state_dict()
    worst = []
    for k in rd:
        if 'running' in k:
            a, b = sd[k].cpu().double(), rd[k].double()
            worst.append((((a - b).norm() / (b.norm() + 1e-30)).item(), k))
    worst.sort(reverse=True)
    print('  running stats worst:', worst[:4])
    mine = net._qat_state.cpu().numpy()
    theirs = qat_observers(ref)
    d = np.abs(mine - theirs) / (np.abs(theirs) + 1e-6)
    d[np.isnan(d)] = 0
    bad = np.argwhere(d > 1e-3)
    print('  observer rows off:', len(set(bad[:, 0].tolist())), 'of', len(mine))
    inv = {v: k for k, v in net._qat_index.items()}
    for r in sorted(set(bad[:, 0].tolist()))[:8]:
        print('   ', inv[r], mine[r], theirs[r])
