"""Sensitivity of the step's wall time to each kernel class: the forward / two-stream backward op lists replayed with ONE class
removed (buffers hold whatever they hold: timing only).  wall(all) - wall(without X) = what the step would gain if X were free,
i.e. how much of X's device time sits on the critical path.
    python scripts/critical_path.py 16 512 [reps]"""
import ctypes as C, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel
import bench

B, S = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
torch.manual_seed(0)
model = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('MODEL', 'mobilenetv2-fpn')))).cuda(); model.train()
x = torch.randn(B, 3, S, S).cuda()
tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, S, 20, seed=1))
plan = model._plan(x, True)
plan._patch(x, tgt)
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
side = model._side_stream(); sd = C.c_void_p(side.cuda_stream)

def wall(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def run2(ops):
    arr = (abi.Op * len(ops))(*ops)
    return wall(lambda: abi.check(l.pq_run_ops2(arr, len(ops), st, sd)))

def run1(ops):
    arr = (abi.Op * len(ops))(*ops)
    return wall(lambda: abi.check(l.pq_run_ops(arr, len(ops), st)))

def cls(op):
    return bench.KERNEL_CLASS.get(op.code, ('other', ''))[0]

for tag, ops in (('fwd', [plan.fwd_arr[k] for k in range(len(plan.fwd))]), ('bwd', [plan.bwd_arr[k] for k in range(len(plan.bwd))])):
    full = run2(ops)
    print('%s: %d ops, two streams %.3f ms, one stream %.3f ms' % (tag, len(ops), full, run1(ops)))
    main = [o for o in ops if not (o.flags & abi.OPF_SIDE)]
    sideo = [o for o in ops if (o.flags & abi.OPF_SIDE)]
    if sideo:
        print('%s: main-stream ops alone %.3f ms (%d), side-stream ops alone %.3f ms (%d)' % (tag, run1(main), len(main), run1(sideo), len(sideo)))
    names = []
    for o in ops:
        key = (cls(o), bool(o.flags & abi.OPF_SIDE))
        if key not in names: names.append(key)
    for name, is_side in names:
        sel = [o for o in ops if (cls(o), bool(o.flags & abi.OPF_SIDE)) == (name, is_side)]
        rest = [o for o in ops if (cls(o), bool(o.flags & abi.OPF_SIDE)) != (name, is_side)]
        alone = run1(sel)
        without = run2(rest)
        print('%s %-66s %-4s %3d launches  alone %.3f ms   step without it %.3f (gain %.3f = %3.0f %% of alone)' % (
            tag, name, 'side' if is_side else 'main', len(sel), alone, without, full - without, 100 * (full - without) / max(alone, 1e-9)))
