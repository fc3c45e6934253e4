"""Per-op table of one op code in a compiled train step (HIP events around every op, replayed as in the step: two streams).
usage: MODEL=regnetx-600m-fpn python scripts/op_table.py <batch> <size> <op name, e.g. ROUTE_BWD | COPY_FWD | BN_BWD_FIN>"""
import ctypes as C, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pqdet_amd import abi
from pqdet_amd.dataset import labels
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel

B, S, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
code = getattr(abi, 'OP_' + name)
model = DetectionModel(io.StringIO(zoo.cfg_text(os.environ.get('MODEL', 'mobilenetv2-fpn')))).cuda(); model.train()
x = torch.randn(B, 3, S, S).cuda()
tgt = tuple(t.cuda() for t in labels.synthetic_targets(B, S, 20, seed=1))
plan = model._plan(x, True)
plan._patch(x, tgt)
l = abi.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sd = C.c_void_p(model._side_stream().cuda_stream)
for arr, n, tag in ((plan.fwd_arr, len(plan.fwd), 'fwd'), (plan.bwd_arr, len(plan.bwd), 'bwd')):
    ms = (C.c_float * n)(); tot = [0.0] * n
    for _ in range(5):
        abi.check(l.pq_run_ops2_timed(arr, n, st, sd, ms))
        for k in range(n): tot[k] += ms[k] / 5
    for k in range(n):
        if arr[k].code != code: continue
        a = getattr(arr[k].u, abi.OP_FIELD[code])
        desc = ''
        if code == abi.OP_ROUTE_BWD:
            desc = 'B%d %dx%d C%d up%d coff%d %s fin%d acc%d act%d stats%d' % (a.B, a.H, a.W, a.C, a.up, a.coff, 'two-sinks' if a.dx2.g else 'one-sink',
                                                                            a.dx.finalize, a.dx.accumulate, a.dx.act, 1 if a.dx.mean else 0)
        elif code == abi.OP_COPY_FWD:
            desc = 'B%d %dx%d C%d up%d coff%d %s' % (a.B, a.H, a.W, a.C, a.up, a.coff, 'a+b' if a.b.x else 'a')
        elif code in (abi.OP_DW_BWD, abi.OP_DW_FWD):
            n_in, n_out = a.B * a.IH * a.IW * a.C, a.B * a.OH * a.OW * a.C
            byt = 4.0 * (2 * n_in + 2 * n_out) if code == abi.OP_DW_BWD else 4.0 * (n_in + n_out)
            desc = 'B%d %dx%d C%d s%d  %.0f MB minimal  %.2f TB/s' % (a.B, a.IH, a.IW, a.C, a.stride, byt / 1e6, byt / (tot[k] * 1e-3) / 1e12 if tot[k] > 0 else 0)
        elif code in (abi.OP_CONV1X1_FWD, abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD):
            fl = 2.0 * a.M * a.I * a.O
            desc = 'M%d I%d O%d  %.1f TF/s' % (a.M, a.I, a.O, fl / (tot[k] * 1e-3) / 1e12 if tot[k] > 0 else 0)
        print('%s op %3d  %7.1f us  %s' % (tag, k, tot[k] * 1e3, desc))
