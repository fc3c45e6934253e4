"""BatchNorm finalize launches left in the step plans of the three BASELINE nets (forward hand-over, backward consumer-side
hand-over, backward producer-side tail) and, for the launches that stay, which op produced their partial rows."""
import io, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
from pqdet_amd import abi
import bench
for name, B, S in (('mobilenetv2-fpn', 16, 512), ('regnetx-600m-fpn', 16, 512), ('regnetx-600m-pan', 8, 960)):
    m = DetectionModel(io.StringIO(zoo.cfg_text(name))).cuda(); m.train()
    x = torch.randn(B, 3, S, S).cuda()
    plan = m._plan(x, True)
    nfin = sum(1 for op in plan.fwd if op.code == abi.OP_BN_FWD_FIN)
    nbfin = [k for k, op in enumerate(plan.bwd) if op.code == abi.OP_BN_BWD_FIN]
    print(name, 'forward: handed over', plan.bn_handover, 'finalize launches left', nfin,
          '| backward: consumer-side', plan.bn_bwd_handover, 'producer tail', getattr(plan, 'bn_bwd_tail', 0), 'finalize launches left', len(nbfin))
    for k in nbfin:
        f = plan.bwd[k].u.bnb
        j = k - 1
        while j >= 0 and (plan.bwd[j].flags & abi.OPF_SIDE or (plan.bwd[j].code == abi.OP_FILL)):
            j -= 1
        print('    bwd#%d C=%d nblk=%d behind %s' % (k, f.C, f.nblk, bench.KERNEL_CLASS.get(plan.bwd[j].code, ('?',))[0]))
    del m, plan
