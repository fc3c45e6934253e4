import io, sys, torch
sys.path.insert(0, '/root/repo')
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
from pqdet_amd.dataset import labels
from pqdet_amd import abi
for name, B, S in (('mobilenetv2-fpn', 16, 512), ('regnetx-600m-fpn', 16, 512), ('regnetx-600m-pan', 8, 960)):
    m = DetectionModel(io.StringIO(zoo.cfg_text(name))).cuda(); m.train()
    x = torch.randn(B, 3, S, S).cuda()
    plan = m._plan(x, True)
    nfin = sum(1 for op in plan.fwd if op.code == abi.OP_BN_FWD_FIN)
    print(name, 'handed over', plan.bn_handover, 'finalize launches left', nfin)
