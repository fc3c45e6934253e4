"""Folded backward of linear shortcuts (engine: Plan._sink / pq_gsink.addend; PQ_SCFOLD=0 switches it off).

mobilenetv2 adds a block's input to the output of its project conv (conv + BatchNorm, no activation on either side; reference
model/parser.py:149-156 ShortCut as mobilenetv2-fpn.cfg uses it).  The gradient of the sum IS the gradient of both addends: the
project conv's gradient tensor is the sum's buffer (its BatchNorm-backward sums come from the sink that finalises the sum's gradient)
and the block input's finalising sink reads its first contribution from there -- no pq_route_bwd launch, no copy.
(Not one of conftest.KERNEL_TEST_MODULES: this module launches whole steps.  Kernel level: the addend cases of test_kernels_gpu.py /
test_irblock_gpu.py.)"""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

from pqdet_amd import abi  # noqa: E402
from pqdet_amd.model import zoo  # noqa: E402
from pqdet_amd.model.interpreter import DetectionModel  # noqa: E402
from seedfill import seeded_fill, seeded_images  # noqa: E402

DEV = 'cuda'


def _step(model_name, B, S):
    from test_model_gpu import _targets
    net = DetectionModel(io.StringIO(zoo.cfg_text(model_name)))
    seeded_fill(net, 4)
    net = net.to(DEV)
    net.train()
    x = seeded_images(B, S, S, 3).to(DEV)
    tgt = tuple(t.to(DEV) for t in _targets(B, S, S, 20, seed=9))
    out = net(x, tgt)
    out['loss'].mean().backward()
    torch.cuda.synchronize()
    net.check_nan()
    plan = list(net._plans.values())[0]
    n_route = sum(1 for op in plan.bwd if op.code == abi.OP_ROUTE_BWD)
    return out['loss'].detach().clone(), net._flat.grad.clone(), plan.sc_folded, n_route, net


@pytest.mark.parametrize('B,S', [(4, 256), (16, 512), (3, 160)])
def test_folded_linear_shortcuts_give_the_gradients_of_the_route_backward_launches(B, S, monkeypatch):
    runs = {}
    for mode in ('1', '0', '1'):
        monkeypatch.setenv('PQ_SCFOLD', mode)
        runs.setdefault(mode, []).append(_step('mobilenetv2-fpn', B, S))
    on, off = runs['1'][0], runs['0'][0]
    assert torch.equal(on[0], off[0])                                    # same forward
    assert torch.equal(on[1], runs['1'][1][1])                           # deterministic
    assert on[2] == 10 and off[2] == 0                                   # the ten residual blocks of mobilenetv2
    assert on[3] == off[3] - 10, (on[3], off[3])                         # ... and their ten pq_route_bwd launches are gone
    flat = on[4]._flat
    gmax = max(off[1][o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
    for p, o in zip(flat.params, flat.offsets):
        da, db = on[1][o:o + p.numel()].double(), off[1][o:o + p.numel()].double()
        # same values through the same arithmetic; only the BatchNorm-backward sums of the project convs are rows of another kernel
        assert (da - db).norm().item() <= 1e-5 * db.norm().item() + 1e-7 * gmax, (o, (da - db).norm().item(), db.norm().item())


@pytest.mark.parametrize('model_name,B,S', [('regnetx-600m-fpn', 2, 256), ('regnetx-600m-pan', 2, 320)])
def test_shortcuts_with_an_activation_share_the_gradient_buffer(model_name, B, S, monkeypatch):
    """RegNet: ReLU behind the sum.  The sum's sinks need the sum's own y, so they cannot emit the conv branch's BatchNorm sums; the
    gradient w.r.t. the raw sum is still the gradient of both addends: the conv branch uses the sum's buffer (one statistics-only
    pq_route_bwd pass, finalize = 3), the other addend reads it as its sink's addend.  Same gradients as the two-sink launches."""
    runs = {}
    for mode in ('1', '0', '1'):
        monkeypatch.setenv('PQ_SCFOLD', mode)
        loss, grad, folded, n_route, net = _step(model_name, B, S)
        plan = list(net._plans.values())[0]
        runs.setdefault(mode, []).append((loss, grad, plan.sc_aliased, n_route, net))
    on, off = runs['1'][0], runs['0'][0]
    assert torch.equal(on[0], off[0]) and torch.equal(on[1], runs['1'][1][1])
    assert on[2] >= 8 and off[2] == 0, (on[2], off[2])
    flat = on[4]._flat
    gmax = max(off[1][o:o + p.numel()].double().norm().item() for p, o in zip(flat.params, flat.offsets))
    for p, o in zip(flat.params, flat.offsets):
        da, db = on[1][o:o + p.numel()].double(), off[1][o:o + p.numel()].double()
        assert (da - db).norm().item() <= 1e-5 * db.norm().item() + 1e-7 * gmax, (o, (da - db).norm().item(), db.norm().item())
