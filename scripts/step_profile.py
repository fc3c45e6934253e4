"""Per-op device time of one train step (HIP events around every op).  usage: step_profile.py B SIZE [passes]"""
import ctypes as C, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi, optim
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
import bench

def log(*a):
    print(*a, file=sys.stderr, flush=True)

B, S = int(sys.argv[1]), int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
torch.manual_seed(0)
model = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('MODEL', 'mobilenetv2-fpn')))).cuda(); model.train()
x = torch.randn(B, 3, S, S).cuda()
tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, S, 20, seed=1))
log('building plan')
plan = model._plan(x, True)
plan._patch(x, tgt)
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rows = []
for tag, arr, n in (('fwd', plan.fwd_arr, len(plan.fwd)), ('bwd', plan.bwd_arr, len(plan.bwd))):
    ms = (C.c_float * n)()
    for p in range(passes):
        t0 = time.time()
        abi.check(l.pq_run_ops_timed(arr, n, st, ms), tag)
        log(tag, 'pass', p, 'wall %.3fs' % (time.time() - t0), 'sum %.3f ms' % sum(ms))
    for k in range(n):
        op = arr[k]
        name = bench.KERNEL_CLASS.get(op.code, ('other', ''))[0]
        fl, by = bench.op_work(op)
        desc = ''
        if op.code == abi.OP_CONV1X1_FWD: a = op.u.c1f; desc = 'M=%d I=%d O=%d' % (a.M, a.I, a.O)
        elif op.code == abi.OP_CONV1X1_DGRAD: a = op.u.c1d; desc = 'M=%d I=%d O=%d' % (a.M, a.I, a.O)
        elif op.code == abi.OP_CONV1X1_WGRAD: a = op.u.c1w; desc = 'M=%d I=%d O=%d' % (a.M, a.I, a.O)
        elif op.code == abi.OP_CONV1X1_BWD: a = op.u.c1b; desc = 'M=%d I=%d O=%d' % (a.M, a.I, a.O)
        elif op.code in (abi.OP_DW_FWD, abi.OP_DW_DGRAD, abi.OP_DW_WGRAD):
            a = {abi.OP_DW_FWD: op.u.dwf, abi.OP_DW_DGRAD: op.u.dwd, abi.OP_DW_WGRAD: op.u.dww}[op.code]
            desc = 'B=%d %dx%d->%dx%d C=%d s=%d' % (a.B, a.IH, a.IW, a.OH, a.OW, a.C, a.stride)
        elif op.code in (abi.OP_GCONV_FWD, abi.OP_GCONV_DGRAD, abi.OP_GCONV_WGRAD):
            a = {abi.OP_GCONV_FWD: op.u.gcf, abi.OP_GCONV_DGRAD: op.u.gcd, abi.OP_GCONV_WGRAD: op.u.gcw}[op.code]
            desc = 'B=%d %dx%d->%dx%d C=%d gw=%d s=%d' % (a.B, a.IH, a.IW, a.OH, a.OW, a.C, a.gw, a.stride)
        rows.append((ms[k], tag, k, name, desc, fl, by))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total device ms (fwd+bwd): %.3f' % tot)
for r in rows[:int(os.environ.get('TOPN', '40'))]:
    ms_, tag, k, name, desc, fl, by = r
    print('%8.3f ms  %s#%-3d %-38s %-34s %7.1f TF/s %8.1f GB/s' % (ms_, tag, k, name, desc, fl / ms_ / 1e9 if ms_ > 0 else 0, by / ms_ / 1e6 if ms_ > 0 else 0))
agg = {}
for r in rows:
    e = agg.setdefault(r[3], [0.0, 0]); e[0] += r[0]; e[1] += 1
print('--- by class')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%9.3f ms %4d launches  %s' % (v[0], v[1], k))
