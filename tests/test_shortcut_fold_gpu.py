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


FLUSH_CFG = """[net]
channels=3

[convolutional]
filters=16
size=3
stride=2
pad=1
batch_normalize=1
activation=relu6

[convolutional]
filters=16
size=3
stride=2
pad=1
groups=16
batch_normalize=1
activation=relu6

[convolutional]
filters=16
size=1
stride=1
pad=1
batch_normalize=1
activation=linear

[convolutional]
filters=16
size=1
stride=1
pad=1
batch_normalize=1
activation=linear

[route]
layers=-2

[convolutional]
filters=24
size=1
stride=2
pad=1
batch_normalize=1
activation=relu6

[convolutional]
filters=75
size=1
stride=1
pad=1
activation=linear

[yolo]
classes=20
ignore_thresh=0.5
bbox_loss=l1
l1_loss_gain=0.05

[route]
layers=3

[shortcut]
from=2
activation=linear

[convolutional]
filters=16
size=3
stride=2
pad=1
groups=16
batch_normalize=1
activation=relu6

[convolutional]
filters=16
size=3
stride=2
pad=1
groups=16
batch_normalize=1
activation=relu6

[convolutional]
filters=75
size=1
stride=1
pad=1
activation=linear

[yolo]
classes=20
ignore_thresh=0.5
bbox_loss=l1
l1_loss_gain=0.05
"""


def test_pending_addend_in_front_of_a_sink_that_cannot_read_it_is_copied():
    """A tensor whose first gradient contribution is a folded shortcut's (pending pq_gsink.addend) and whose NEXT consumer in the
    backward is a strided 1x1 conv -- the one sink path that is assembled by hand and cannot read an addend: the engine materialises
    the addend with one pq_route_bwd copy (Plan._flush_addend) and carries on.  No BASELINE net has this topology; the gradients are
    checked against the CPU oracle with frozen BatchNorm statistics (well conditioned), 5e-3 norm-wise like the pruned-net test."""
    from oracle.net import OracleNet
    from pqdet_amd import tools
    from test_model_gpu import _targets, _relerr
    net = DetectionModel(io.StringIO(FLUSH_CFG))
    seeded_fill(net, 31)
    ref = OracleNet(io.StringIO(FLUSH_CFG))
    ref.load_state_dict(net.state_dict())
    net = net.to(DEV)
    x = seeded_images(2, 128, 160, 6)
    tgt = _targets(2, 128, 160, 20, seed=4)
    net.train(), ref.eval()
    tools.freeze_bn_stats(net)
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    net.check_nan()
    plan = list(net._plans.values())[0]
    assert plan.sc_folded == 1
    copies = [op for op in plan.bwd if op.code == abi.OP_ROUTE_BWD and not op.u.rb.dx.accumulate and not op.u.rb.dx.finalize and not op.u.rb.dx.addend]
    assert len(copies) == 1, 'expected exactly one materialising copy of a pending addend'
    want = ref(x, tgt)
    want['loss'].mean().backward()
    for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss'):
        assert abs(out[k].item() - want[k].item()) <= 1e-3 * abs(want[k].item()) + 1e-5, k
    for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
        if p2.grad.norm().item() > 1e-6:
            assert _relerr(p1.grad, p2.grad) < 5e-3, n1
