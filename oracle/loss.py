"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU restatement of the reference YOLO loss and IoU family.

* ``pair_iou`` / ``giou`` / ``diou`` / ``ciou``  -- reference ``tools.py:357-477`` (torch versions, *no epsilon*;
  ``diou`` and ``ciou`` reproduce the as-written sign: ``+ d_centre^2 / d_enclose^2`` at :437 / :477, and ciou's
  ``alpha`` is taken outside the autograd graph, :473-475)
* ``iou_xyxy_np`` -- reference ``tools.py:335-355`` (``iou_calc1``: union clamped at 1e-14)
* ``iou_xywh_np`` -- reference ``tools.py:479-505``
* ``loss_per_scale`` -- reference ``model/loss.py:7-115``; hard-coded gains/focal constants at :35-41
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _areas_inter(b1, b2):
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    lo = torch.maximum(b1[..., :2], b2[..., :2])
    hi = torch.minimum(b1[..., 2:], b2[..., 2:])
    wh = (hi - lo).clamp_min(0)
    inter = wh[..., 0] * wh[..., 1]
    return a1, a2, inter


def pair_iou(b1, b2):
    """tools.py:357-376 ``iou_calc3``."""
    a1, a2, inter = _areas_inter(b1, b2)
    return inter / (a1 + a2 - inter)


def _giou_parts(b1, b2):
    a1, a2, inter = _areas_inter(b1, b2)
    union = a1 + a2 - inter
    iou = inter / union
    elo = torch.minimum(b1[..., :2], b2[..., :2])
    ehi = torch.maximum(b1[..., 2:], b2[..., 2:])
    ewh = (ehi - elo).clamp_min(0)
    earea = ewh[..., 0] * ewh[..., 1]
    return iou, iou - (earea - union) / earea, elo, ehi


def giou(b1, b2):
    """tools.py:378-404."""
    return _giou_parts(b1, b2)[1]


def _centre_term(b1, b2, elo, ehi):
    c1 = (b1[..., :2] + b1[..., 2:]) / 2
    c2 = (b2[..., :2] + b2[..., 2:]) / 2
    return (c1 - c2).pow(2).sum(-1) / (elo - ehi).pow(2).sum(-1)


def diou(b1, b2):
    """tools.py:406-437 (note the '+', as written)."""
    _, g, elo, ehi = _giou_parts(b1, b2)
    return g + _centre_term(b1, b2, elo, ehi)


def ciou(b1, b2):
    """tools.py:439-477."""
    iou, g, elo, ehi = _giou_parts(b1, b2)
    w1, h1 = b1[..., 2] - b1[..., 0], b1[..., 3] - b1[..., 1]
    w2, h2 = b2[..., 2] - b2[..., 0], b2[..., 3] - b2[..., 1]
    v = (4 / (math.pi ** 2)) * (torch.atan(w1 / h1) - torch.atan(w2 / h2)).pow(2)
    with torch.no_grad():
        alpha = v / ((1 - iou) + v)
    return g + _centre_term(b1, b2, elo, ehi) + alpha * v


def iou_xyxy_np(b1: np.ndarray, b2: np.ndarray):
    """tools.py:335-355 ``iou_calc1``."""
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    wh = np.maximum(np.minimum(b1[..., 2:], b2[..., 2:]) - np.maximum(b1[..., :2], b2[..., :2]), 0.0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / np.maximum(a1 + a2 - inter, 1e-14)


def iou_xywh_np(b1: np.ndarray, b2: np.ndarray):
    """tools.py:479-505 (centre-size boxes)."""
    a1 = b1[..., 2] * b1[..., 3]
    a2 = b2[..., 2] * b2[..., 3]
    c1 = np.concatenate([b1[..., :2] - b1[..., 2:] * 0.5, b1[..., :2] + b1[..., 2:] * 0.5], -1)
    c2 = np.concatenate([b2[..., :2] - b2[..., 2:] * 0.5, b2[..., :2] + b2[..., 2:] * 0.5], -1)
    wh = np.maximum(np.minimum(c1[..., 2:], c2[..., 2:]) - np.maximum(c1[..., :2], c2[..., :2]), 0.0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (a1 + a2 - inter)


def _focal(target, actual, alpha, gamma):
    """model/loss.py:17-20."""
    return 2 * torch.abs(target - 1 + alpha) * torch.pow(torch.abs(target - actual), gamma)


def _smooth_l1_mean(pred, tgt, beta=1 / 9):
    """model/loss.py:7-15."""
    n = torch.abs(pred - tgt)
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta).mean(-1, keepdim=True)


_BOX_FN = {'giou': giou, 'diou': diou, 'ciou': ciou, 'iou': pair_iou}


def loss_per_scale(pred, label, bboxes, opt, check_nan=True):
    """model/loss.py:22-115.  pred (B,H,W,A,5+C) decoded; label (B,H,W,A,6+C); bboxes (B,M,4)."""
    stride, kind, thresh = opt['stride'], opt['bbox_loss'], opt['ignore_thresh']
    h, w = pred.shape[1:3]
    in_area = (stride * h) * (stride * w)
    p_box, p_conf, p_cls = pred[..., 0:4], pred[..., 4:5], pred[..., 5:]
    t_box, respond, t_cls, mixw = label[..., 0:4], label[..., 4:5], label[..., 5:-1], label[..., -1:]

    twh = t_box[..., 2:] - t_box[..., :2]
    scale = 2.0 - 1.0 * twh[..., 0:1] * twh[..., 1:2] / in_area
    if kind == 'l1':
        box = respond * scale * _smooth_l1_mean(p_box, t_box) * opt['l1_loss_gain']
    elif kind in _BOX_FN:
        box = respond * scale * (1.0 - _BOX_FN[kind](p_box, t_box)[..., None])
    else:
        raise NotImplementedError
    box = box * 1                                                     # bbox_loss_gain

    iou = pair_iou(p_box[:, :, :, :, None, :], bboxes[:, None, None, None, :, :])
    best = iou.max(-1)[0][..., None]
    background = (1.0 - respond) * (best < thresh).float()
    bce_conf = F.binary_cross_entropy(p_conf, respond, reduction='none')
    conf = 1 * _focal(respond, p_conf, 0.75, 2) * (respond * bce_conf + background * bce_conf)
    cls = 2 * _focal(t_cls, p_cls, 0.5, 2) * respond * F.binary_cross_entropy(p_cls, t_cls, reduction='none')

    def per_batch(t):
        return (t * mixw).sum([1, 2, 3, 4]).mean(0, keepdim=True)

    box, conf, cls = per_batch(box), per_batch(conf), per_batch(cls)
    total = box + conf + cls
    if check_nan and torch.isnan(total):
        raise RuntimeError('NaN in loss')
    return total, box, conf, cls
