"""Wall time of the forward and backward op lists of one train step, replayed back to back (no per-op events)."""
import ctypes as C, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel

B, S = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
torch.manual_seed(0)
if os.environ.get('OWN_STREAM'):          # a non-default stream (stream capture refuses the legacy default stream)
    torch.cuda.set_stream(torch.cuda.Stream())
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

nf, nb = len(plan.fwd), len(plan.bwd)
print('ops: fwd %d bwd %d' % (nf, nb))
print('forward            %.3f ms' % wall(lambda: abi.check(l.pq_run_ops(plan.fwd_arr, nf, st))))
if getattr(plan, 'fwd_two_streams', False):
    print('forward 2 streams  %.3f ms  (side layers %s)' % (wall(lambda: abi.check(l.pq_run_ops2(plan.fwd_arr, nf, st, sd))), plan.fwd_side_layers))
print('backward 2 streams %.3f ms' % wall(lambda: abi.check(l.pq_run_ops2(plan.bwd_arr, nb, st, sd))))
print('backward 1 stream  %.3f ms' % wall(lambda: abi.check(l.pq_run_ops(plan.bwd_arr, nb, st))))
# per op class, back to back: replay only the ops of one class (inputs are whatever the buffers hold)
import bench
for arr, n, tag in ((plan.fwd_arr, nf, 'fwd'), (plan.bwd_arr, nb, 'bwd')):
    classes = {}
    for k in range(n):
        classes.setdefault(bench.KERNEL_CLASS.get(arr[k].code, ('other', ''))[0], []).append(k)
    for name, idxs in classes.items():
        sub = (abi.Op * len(idxs))(*[arr[k] for k in idxs])
        ms = wall(lambda: abi.check(l.pq_run_ops(sub, len(idxs), st)))
        fl = sum(bench.op_work(arr[k])[0] for k in idxs); by = sum(bench.op_work(arr[k])[1] for k in idxs)
        print('%s %-60s %4d launches %.3f ms  (%.1f us each)  %6.1f TF/s %7.0f GB/s (algorithmic)' % (
            tag, name, len(idxs), ms, ms / len(idxs) * 1e3, fl / ms / 1e9, by / ms / 1e6))

# per GEMM op, replayed alone (hot caches: optimistic for the big layers): time against max(MFMA, HBM) bound
if os.environ.get('PER_OP'):
    rows = []
    for arr, n, tag in ((plan.fwd_arr, nf, 'fwd'), (plan.bwd_arr, nb, 'bwd')):
        for k in range(n):
            code = arr[k].code
            if code not in (abi.OP_CONV1X1_FWD, abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD, abi.OP_CONV1X1_BWD):
                continue
            a = getattr(arr[k].u, abi.OP_FIELD[code])
            sub = (abi.Op * 1)(arr[k])
            ms = wall(lambda: abi.check(l.pq_run_ops(sub, 1, st)))
            fl = 2.0 * a.M * a.I * a.O * (2 if code == abi.OP_CONV1X1_BWD else 1)
            by = 4.0 * a.M * {abi.OP_CONV1X1_FWD: a.I + a.O, abi.OP_CONV1X1_DGRAD: 2 * a.O + 2 * a.I,
                              abi.OP_CONV1X1_WGRAD: 2 * a.O + a.I, abi.OP_CONV1X1_BWD: 2 * a.O + 3 * a.I}[code]
            bound = max(fl / 157.3e9, by / 4.0e9)          # ms
            rows.append((ms - bound, ms, bound, tag, bench.KERNEL_CLASS[code][0][:14], a.M, a.I, a.O))
    agg = {}
    for r in rows:
        key = r[4:]
        e = agg.setdefault(key, [0, 0.0, 0.0]); e[0] += 1; e[1] += r[1]; e[2] += r[2]
    print('--- GEMM ops alone: count, total ms, bound ms (max of fp32 MFMA peak and real bytes at 4 TB/s), excess')
    for key, e in sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:40]:
        print('%-14s M=%-8d I=%-5d O=%-5d x%d  %.3f ms  bound %.3f  excess %.3f' % (key + (e[0], e[1], e[2], e[1] - e[2])))
    print('total %.3f ms, bound %.3f ms' % (sum(e[1] for e in agg.values()), sum(e[2] for e in agg.values())))

# every op alone (hot caches), biggest first: name, shape hint, us, algorithmic GB/s
if os.environ.get('PER_OP_ALL'):
    rows = []
    for arr, n, tag in ((plan.fwd_arr, nf, 'fwd'), (plan.bwd_arr, nb, 'bwd')):
        for k in range(n):
            code = arr[k].code
            if os.environ.get('ONLY') and os.environ['ONLY'] not in bench.KERNEL_CLASS.get(code, ('other',))[0]:
                continue
            a = getattr(arr[k].u, abi.OP_FIELD[code])
            sub = (abi.Op * 1)(arr[k])
            ms = wall(lambda: abi.check(l.pq_run_ops(sub, 1, st)))
            fl, by = bench.op_work(arr[k])
            hint = ' '.join('%s=%s' % (f, getattr(a, f)) for f in ('M', 'I', 'O', 'B', 'IH', 'IW', 'OH', 'OW', 'H', 'W', 'C', 'stride', 'part', 'up')
                            if hasattr(a, f))
            rows.append((ms, tag, k, bench.KERNEL_CLASS.get(code, ('other',))[0][:28], hint, by, fl))
    print('--- every op alone, >= 15 us')
    for ms, tag, k, name, hint, by, fl in sorted(rows, key=lambda r: -r[0]):
        if ms * 1e3 < float(os.environ.get('MIN_US', 15)): break
        print('%s#%-3d %-28s %-70s %7.1f us %6.0f GB/s %5.1f TF' % (tag, k, name, hint, ms * 1e3, by / ms / 1e6, fl / ms / 1e9))
