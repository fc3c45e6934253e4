"""YOLO loss entry point -- same signature and return contract as the reference ``model/loss.py:22-115``.

``loss_per_scale(pred, label, bboxes, opt) -> (loss, bbox_loss, conf_loss, prob_loss)``, each of shape ``(1,)``,
differentiable w.r.t. ``pred``.  The arithmetic (box term in the five ``bbox_loss`` modes, focal-weighted BCE objectness
with the IoU ignore mask, focal BCE class term, mix-up weights, per-batch mean) runs in the fused HIP kernels
``pq_yolo_loss_fwd`` / ``pq_yolo_loss_bwd``; the constants the reference hard-codes at :35-41 are hard-coded there too.

NaN handling: the reference calls ``torch.isnan(loss)`` (a host sync) and raises ``RuntimeError('NaN in loss')``
(:110-114).  This standalone entry keeps that behaviour.  The engine path (``DetectionModel.forward``) uses the same kernel
but polls the device flag one step late to keep the training loop asynchronous ("deliberate deviations").
"""
import torch

from .. import abi, ops


def loss_per_scale(pred: torch.Tensor, label: torch.Tensor, bboxes: torch.Tensor, opt: dict):
    kind = opt['bbox_loss']
    if kind not in abi.BOX_KIND:
        raise NotImplementedError
    total, box, conf, cls, nan_flag = ops.yolo_loss(pred, label, bboxes, opt['stride'], abi.BOX_KIND[kind],
                                                    opt['ignore_thresh'], opt['l1_loss_gain'])
    if bool(nan_flag.item()):
        print('xy: {}, conf: {}, cls: {}'.format(box.item(), conf.item(), cls.item()))
        raise RuntimeError('NaN in loss')
    return total, box, conf, cls
