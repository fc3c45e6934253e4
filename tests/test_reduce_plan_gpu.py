"""The engine's use of pq_reduce_batch (PQ_REDUCE_BATCH=1): plan-level checks -- gradients against the per-layer reductions, launch
sets, validity of the gradient-ready marks a DDP reducer is offered.  (Kernel-level tests: test_reduce_batch_gpu.py; this module
is deliberately NOT one of conftest.KERNEL_TEST_MODULES: it launches whole steps.)
Work mirrored: the weight gradients torch.autograd accumulates behind the reference's trainer.py:231-233, averaged per bucket like
nn.DataParallel does at tools.py:215-216."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from pqdet_amd import abi  # noqa: E402

DEV = 'cuda'

def _train_step(model_name, B, S, recorder=None):
    import io
    from pqdet_amd.model import zoo
    from pqdet_amd.model.interpreter import DetectionModel
    from test_model_gpu import _targets
    from seedfill import seeded_fill, seeded_images
    net = DetectionModel(io.StringIO(zoo.cfg_text(model_name)))
    seeded_fill(net, 4)
    net = net.to(DEV)
    net.train()
    net._reducer = recorder
    x = seeded_images(B, S, S, 3).to(DEV)
    tgt = tuple(t.to(DEV) for t in _targets(B, S, S, 20, seed=9))
    out = net(x, tgt)
    abi.lib().pq_launch_log(1)
    out['loss'].mean().backward()
    torch.cuda.synchronize()
    from conftest import dump_launch_log
    launches = dump_launch_log()
    abi.lib().pq_launch_log(0)
    net.check_nan()
    plan = list(net._plans.values())[0]
    return out['loss'].detach().clone(), net._flat.grad.clone(), plan, net, launches


class MarkRecorder:
    """Stands in for parallel.GradReducer: EVERY gradient-ready mark the plan offers closes a bucket, and the range it declares final
    is copied at that moment (after both compute streams have drained)."""
    def __init__(self):
        self.snaps = []

    def pick_marks(self, segments, total):
        marks, hi = [], total
        for op_end, lo in segments:
            if lo < hi:
                marks.append((op_end, lo))
                hi = lo
        return marks

    def reduce_range(self, flat, lo, hi, also_wait=None):
        torch.cuda.synchronize()
        self.snaps.append((lo, hi, flat[lo:hi].clone()))

    def finish(self):
        pass


@pytest.mark.parametrize('model_name,B,S', [('mobilenetv2-fpn', 4, 256), ('regnetx-600m-fpn', 2, 256)])
def test_plan_with_batched_reductions_equals_plan_with_per_layer_reductions(model_name, B, S, monkeypatch):
    """PQ_REDUCE_BATCH=1: the weight-gradient entries of the step defer their reductions and a handful of pq_reduce_batch launches sum
    the slabs; every parameter gradient agrees with the per-layer reductions to 1e-5 (same slabs, another fixed order), the
    replay is bit-identical, no per-layer reduce kernel is launched any more, and every gradient-ready mark the plan keeps is valid:
    the range it declares final does not change afterwards."""
    runs = {}
    for mode in ('1', '0', '1'):
        monkeypatch.setenv('PQ_REDUCE_BATCH', mode)
        runs.setdefault(mode, []).append(_train_step(model_name, B, S))
    on, off = runs['1'][0], runs['0'][0]
    assert torch.equal(on[0], off[0])                                    # same forward
    assert torch.equal(on[1], runs['1'][1][1])                           # deterministic
    assert 1 <= on[2].red_batches <= 7 and off[2].red_batches == 0
    per_layer = ('slab_reduce', 'dww_partial_reduce', 'dww_slice_rows', 'pw_slab_reduce', 'gconv_wgrad_reduce')
    assert not [e for e in on[4] if any(k in e for k in per_layer)], 'per-layer reduce launches left in the batched plan'
    assert [e for e in off[4] if any(k in e for k in per_layer)] and any('reduce_batch_kernel' in e for e in on[4])
    flat = on[3]._flat
    gmax = max(off[1][o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
    for p, o in zip(flat.params, flat.offsets):
        da, db = on[1][o:o + p.numel()].double(), off[1][o:o + p.numel()].double()
        assert (da - db).norm().item() <= 1e-5 * db.norm().item() + 1e-7 * gmax, (o, (da - db).norm().item(), db.norm().item())
    # marks: bucketed replay with a recorder in the reducer's place
    for mode in ('1', '0'):
        monkeypatch.setenv('PQ_REDUCE_BATCH', mode)
        rec = MarkRecorder()
        _, grad, plan, _, _ = _train_step(model_name, B, S, recorder=rec)
        assert len(rec.snaps) >= (3 if mode == '1' else 10), len(rec.snaps)
        for lo, hi, snap in rec.snaps:
            assert torch.equal(snap, grad[lo:hi]), 'mode %s: range [%d, %d) changed after its mark' % (mode, lo, hi)
        assert torch.equal(grad, runs[mode][0][1])                       # bucketed replay == one-call replay
