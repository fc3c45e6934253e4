"""Which kernels read the p / q / r vectors of every remaining pq_bn_bwd_finalize of a step plan (consumer-side hand-over census)."""
import ctypes as C, io, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pqdet_amd import abi
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
name = sys.argv[1] if len(sys.argv) > 1 else 'mobilenetv2-fpn'
B, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 512)
m = DetectionModel(io.StringIO(zoo.cfg_text(name))).cuda(); m.train()
plan = m._plan(torch.randn(B, 3, S, S).cuda(), True)
l = abi.lib()
DY = {abi.OP_CONV1X1_DGRAD: 'dy', abi.OP_CONV1X1_WGRAD: 'dy', abi.OP_CONV1X1_BWD: 'dy', abi.OP_DW_BWD: 'dy', abi.OP_GCONV_DGRAD: 'dy',
      abi.OP_GCONV_WGRAD: 'dy', abi.OP_DW_DGRAD: 'dy', abi.OP_DW_WGRAD: 'dy', abi.OP_STEM_WGRAD: 'dy'}
def fam(op):
    a = getattr(op.u, abi.OP_FIELD[op.code])
    pl = abi.GemmPlan()
    if op.code == abi.OP_CONV1X1_DGRAD and l.pq_conv1x1_dgrad_plan(C.byref(a), C.byref(pl)) == 0:
        return 'dgrad f%d %dx%d M=%d I=%d O=%d' % (pl.family, pl.bm, pl.bn, a.M, a.I, a.O)
    if op.code == abi.OP_CONV1X1_WGRAD and l.pq_conv1x1_wgrad_plan(C.byref(a), C.byref(pl)) == 0:
        return 'wgrad f%d %dx%d M=%d I=%d O=%d' % (pl.family, pl.bm, pl.bn, a.M, a.I, a.O)
    return {abi.OP_CONV1X1_BWD: 'pw_bwd', abi.OP_DW_BWD: 'dw_bwd(part %d)' % getattr(a, 'part', -1), abi.OP_STEM_WGRAD: 'stem_wgrad'}.get(op.code, 'op%d' % op.code)
cnt = collections.Counter()
for k, op in enumerate(plan.bwd):
    if op.code != abi.OP_BN_BWD_FIN:
        continue
    f = op.u.bnb
    readers = []
    for j in range(k + 1, len(plan.bwd)):
        o2 = plan.bwd[j]
        if o2.code in DY:
            a = getattr(o2.u, abi.OP_FIELD[o2.code])
            if getattr(a, 'dy').p == f.p:
                readers.append(fam(o2))
    key = ' + '.join(sorted(r.split(' M=')[0] for r in readers))
    cnt[key] += 1
    print('bwd#%-3d C=%-4d nblk=%-4d readers: %s' % (k, f.C, f.nblk, ' | '.join(readers)))
print('---')
for k, v in cnt.most_common():
    print('%3d  %s' % (v, k))
