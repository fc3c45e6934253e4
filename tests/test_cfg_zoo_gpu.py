"""Every detector cfg the reference ships (rebuilt as text from the section dicts recorded in fixture G1 -- the files
themselves do not travel) through one training step and one inference on the HIP engine, against the CPU oracle.
Covers the darknet / rsod / visdrone / coco variants, the pruned myolo nets, regnetx fpn / pan / rpan / yolo and
regnety-400m-fpn (squeeze-and-excitation blocks: avgpool, logistic gate, scale_channels)."""
import io
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import postprocess as opost  # noqa: E402
from oracle.net import OracleNet  # noqa: E402
from pqdet_amd.model.interpreter import AnyModel, DetectionModel  # noqa: E402
from seedfill import seeded_boxes, seeded_fill, seeded_images  # noqa: E402

DEV = 'cuda'
HERE = os.path.dirname(os.path.abspath(__file__))
G1 = json.load(open(os.path.join(HERE, 'golden', 'g1_parser.json')))['cfgs']

# every detector cfg; the two myolo-prune30 nets have arbitrary (pruned) channel counts -- 25, 95, 142, 143, ... -- that
# exercise the masked channel-quad paths of the depthwise / elementwise kernels and unaligned concatenations.
# The four classifier cfgs (7 x 7 stem, maxpool, dense 3 x 3 convs, global avgpool, fc) run forward only, like the reference
# itself (AnyModel has no training caller): test_reference_classifier_cfg_infers_like_the_reference below.
UNSUPPORTED = set()
DETECTORS = sorted(k for k in G1 if not k.startswith('classifier/') and k not in UNSUPPORTED)


def cfg_text_from_specs(specs):
    lines = []
    for sec in specs:
        lines.append('[%s]' % sec['name'])
        for k, v in sec.items():
            if k == 'name':
                continue
            if isinstance(v, list):
                v = ', '.join(str(x) for x in v)
            lines.append('%s=%s' % (k, v))
        lines.append('')
    return '\n'.join(lines) + '\n'


def test_the_list_covers_every_reference_detector():
    assert len(DETECTORS) == 22


@pytest.mark.parametrize('rel', DETECTORS)
def test_reference_cfg_trains_and_infers_like_the_oracle(rel):
    text = cfg_text_from_specs(G1[rel]['specs'])
    net = DetectionModel(io.StringIO(text))
    assert len(net.module_list) == G1[rel]['n_layers']
    assert sum(p.numel() for p in net.parameters()) == G1[rel]['params']
    seeded_fill(net, 2)
    ref = OracleNet(io.StringIO(text))
    ref.load_state_dict(net.state_dict())
    net = net.to(DEV)
    ncls = [m.opt['classes'] for m in net.module_list if m._type == 'yolo'][0]
    B, S = 2, 128
    x = seeded_images(B, S, S, 3)
    hw = np.array([[S // s, S // s] for s in (8, 16, 32)])
    per = [opost.create_label(bb, hw, ncls) for bb in seeded_boxes(B, S, ncls, 4)]
    tgt = tuple([torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
                + [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)])
    net.train(), ref.train()
    try:
        want = ref(x, tgt)
    except RuntimeError as e:                       # e.g. bbox_loss=ciou: the reference raises 'NaN in loss' (G3)
        with pytest.raises(RuntimeError, match=str(e)):
            net.nan_check = 'sync'
            net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
        return
    out = net(x.to(DEV), tuple(t.to(DEV) for t in tgt))
    out['loss'].mean().backward()
    net.check_nan()
    for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss'):
        assert abs(out[k].item() - want[k].item()) <= 2e-3 * abs(want[k].item()) + 1e-4, (k, out[k].item(), want[k].item())
    assert torch.isfinite(net.flat_buffers()[1]).all()
    net.eval(), ref.eval()
    with torch.no_grad():
        a, b = net(x.to(DEV)).cpu(), ref(x)
    assert a.shape == b.shape
    assert ((a - b).norm() / b.norm()).item() < 2e-3


CLASSIFIERS = sorted(k for k in G1 if k.startswith('classifier/'))
G13 = np.load(os.path.join(HERE, 'golden', 'golden_r3b.npz'))


def test_the_list_covers_every_reference_classifier():
    assert len(CLASSIFIERS) == 4


@pytest.mark.parametrize('rel', CLASSIFIERS)
def test_reference_classifier_cfg_infers_like_the_reference(rel):
    """Fixture G13 (tests/golden/make_golden_r3b.py: the reference's AnyModel on one seeded 64 x 64 image, seeded fill): the (1000,)
    output and, for the ResNets, the maxpool layer's output; then a non-square input against the CPU oracle."""
    name = os.path.splitext(os.path.basename(rel))[0]
    text = cfg_text_from_specs(G1[rel]['specs'])
    net = AnyModel(io.StringIO(text))
    assert len(net.module_list) == G1[rel]['n_layers']
    assert sum(p.numel() for p in net.parameters()) == G1[rel]['params']
    seeded_fill(net, 5)
    ref = OracleNet(io.StringIO(text))
    ref.load_state_dict(net.state_dict())
    net = net.to(DEV).eval()
    ref.eval()
    with torch.no_grad():
        y = net(seeded_images(1, 64, 64, 7).to(DEV)).cpu().numpy()
    want = G13['g13_%s_out' % name]
    assert y.shape == want.shape == (1000,)
    assert np.linalg.norm(y - want) / np.linalg.norm(want) < 2e-3
    x2 = seeded_images(1, 96, 160, 11)
    with torch.no_grad():
        a, b = net(x2.to(DEV)).cpu(), ref(x2)
    assert a.shape == b.shape == (1000,)
    assert ((a - b).norm() / b.norm()).item() < 2e-3
    # no training path (the reference has none either: no loss without yolo layers)
    net.train()
    with pytest.raises(NotImplementedError):
        net(x2.to(DEV), tuple(torch.zeros(1, device=DEV) for _ in range(6)))


def test_tanh_and_logistic_as_conv_activations_forward():
    """reference model/parser.py:106-112: any ACTIVATION_MAP entry may follow a conv.  No shipped cfg uses tanh / logistic outside
    the squeeze-and-excitation gates; the engine materialises act(bn(conv)) once for them (forward only).  Checked end to end
    on a headless net against the CPU oracle (plain PyTorch modules)."""
    text = """[net]
channels=3

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=relu

[convolutional]
batch_normalize=1
filters=24
size=1
stride=1
pad=1
activation=tanh

[convolutional]
batch_normalize=1
filters=24
size=3
stride=1
pad=1
groups=24
activation=logistic

[convolutional]
batch_normalize=0
filters=32
size=1
stride=1
pad=1
activation=logistic

[maxpool]
size=2
stride=2
pad=0
padding=0

[convolutional]
batch_normalize=1
filters=40
size=3
stride=1
pad=1
activation=leaky

[avgpool]
height=2
width=3
"""
    with pytest.raises(NotImplementedError):          # a depthwise conv with a bias has no kernel input for it: refused, never dropped
        bad = AnyModel(io.StringIO(text.replace("batch_normalize=1\nfilters=24\nsize=3", "batch_normalize=0\nfilters=24\nsize=3")))
        bad.to(DEV).eval()(seeded_images(1, 64, 64, 1).to(DEV))
    net = AnyModel(io.StringIO(text))
    seeded_fill(net, 9)
    ref = OracleNet(io.StringIO(text))
    ref.load_state_dict(net.state_dict())
    net = net.to(DEV).eval()
    ref.eval()
    x = seeded_images(2, 64, 96, 3)
    with torch.no_grad():
        a, b = net(x.to(DEV)).cpu(), ref(x)
    assert a.shape == b.shape == (2, 40, 2, 3)
    assert ((a - b).norm() / b.norm()).item() < 2e-3
