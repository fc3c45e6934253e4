"""Round-2 fixtures recorded from the reference (tests/golden/make_golden_r2.py): well-conditioned end-to-end gradients
(G5b: ``model.eval()`` + targets, 512x512 batch 2), NaN-free CIoU (G4b).  The CPU oracle is pinned against them here
(``-m "not gpu"``); the HIP engine meets the SAME reference vectors at 5e-3 norm-wise (``-m gpu``) -- no 10-25 % bands."""
import io

import numpy as np
import pytest
import torch

from oracle import loss as oloss
from oracle import postprocess as opost
from oracle.net import OracleNet
from pqdet_amd.model import zoo
from seedfill import seeded_boxes, seeded_fill, seeded_images

CFG = {'mbv2': 'mobilenetv2-fpn', 'rxfpn': 'regnetx-600m-fpn', 'rxpan': 'regnetx-600m-pan'}


def _targets(batch, size, ncls, seed):
    hw = np.array([[size // s, size // s] for s in (8, 16, 32)])
    per = [opost.create_label(bb, hw, ncls) for bb in seeded_boxes(batch, size, ncls, seed)]
    return tuple([torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
                 + [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)])


def _relerr(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _check_g5b(net, out, golden_r2, r2_names, tag, tol_loss, tol_grad):
    got = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                   + [v.item() for v in out['loss_per_branch']])
    np.testing.assert_allclose(got, golden_r2['g5b_%s_loss' % tag], rtol=tol_loss)
    params = dict(net.named_parameters())
    assert list(params) == r2_names['g5b'][tag]
    want_norm = golden_r2['g5b_%s_gradnorm' % tag]
    gn = np.array([p.grad.double().norm().item() for p in params.values()])
    scale = want_norm.max()
    np.testing.assert_allclose(gn, want_norm, rtol=tol_grad, atol=1e-6 * scale)
    full = [k for k in golden_r2.files if k.startswith('g5b_%s_grad_' % tag)]
    assert len(full) >= 6
    for key in full:
        name = key[len('g5b_%s_grad_' % tag):]
        assert _relerr(params[name].grad, golden_r2[key]) < tol_grad, name


def _fixture(tag, golden_r2, r2_names, golden_r3, r3_names):
    """G5b of the two round-2 nets lives in golden_r2.npz; regnetx-600m-pan (round 3, make_golden_r3.py: 16 full gradient
    tensors incl. the strided projection, both stride-2 grouped convs of the bottom-up path and the conv behind every
    two-input route) in golden_r3.npz."""
    return (golden_r3, r3_names) if tag == 'rxpan' else (golden_r2, r2_names)


@pytest.mark.parametrize('tag', ['mbv2', 'rxfpn', 'rxpan'])
def test_oracle_meets_reference_eval_bn_gradients(golden_r2, r2_names, golden_r3, r3_names, tag):
    golden_r2, r2_names = _fixture(tag, golden_r2, r2_names, golden_r3, r3_names)
    torch.set_num_threads(8)
    net = OracleNet(io.StringIO(zoo.cfg_text(CFG[tag])))
    seeded_fill(net, 5)
    net.eval()
    out = net(seeded_images(2, 512, 512, 5), _targets(2, 512, 20, 6))
    out['loss'].mean().backward()
    _check_g5b(net, out, golden_r2, r2_names, tag, 1e-5, 2e-4)


def test_oracle_ciou_without_nan(golden_r2):
    b1, b2 = torch.from_numpy(golden_r2['g4b_b1']), torch.from_numpy(golden_r2['g4b_b2'])
    for name in ('giou', 'diou', 'ciou'):
        np.testing.assert_allclose(getattr(oloss, name)(b1, b2).numpy(), golden_r2['g4b_' + name], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(getattr(oloss, name)(b1[:, None, :], b2[None, :, :]).numpy(), golden_r2['g4b_bc_' + name],
                                   rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['mbv2', 'rxfpn', 'rxpan'])
def test_hip_meets_reference_eval_bn_gradients(golden_r2, r2_names, golden_r3, r3_names, tag):
    """The reference's own gradients (BatchNorm on running statistics: no batch-statistics amplification of fp32 round-off)
    at 5e-3 norm-wise for every stored tensor and every parameter norm; losses at 1e-3 (north_star)."""
    from pqdet_amd.model.interpreter import DetectionModel
    golden_r2, r2_names = _fixture(tag, golden_r2, r2_names, golden_r3, r3_names)
    net = DetectionModel(io.StringIO(zoo.cfg_text(CFG[tag])))
    seeded_fill(net, 5)
    net = net.to('cuda')
    net.eval()
    out = net(seeded_images(2, 512, 512, 5).cuda(), tuple(t.cuda() for t in _targets(2, 512, 20, 6)))
    out['loss'].mean().backward()
    _check_g5b(net, out, golden_r2, r2_names, tag, 1e-3, 5e-3)


@pytest.mark.gpu
def test_hip_iou_family_incl_ciou(golden_r2):
    from pqdet_amd import tools
    b1, b2 = torch.from_numpy(golden_r2['g4b_b1']).cuda(), torch.from_numpy(golden_r2['g4b_b2']).cuda()
    for name in ('iou_calc3', 'giou', 'diou', 'ciou'):
        got = getattr(tools, name)(b1[:, None, :], b2[None, :, :]).cpu().numpy()
        np.testing.assert_allclose(got, golden_r2['g4b_bc_' + name], rtol=1e-5, atol=1e-6, err_msg=name)
        got1 = getattr(tools, name)(b1, b2).cpu().numpy()
        np.testing.assert_allclose(got1, golden_r2['g4b_' + name], rtol=1e-5, atol=1e-6, err_msg=name)
