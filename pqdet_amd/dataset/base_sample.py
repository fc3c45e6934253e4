"""Box recovery (undo the eval letterbox) -- mirrors the reference ``dataset/base_sample.py:98-139`` and the per-dataset
affine helpers (``voc_sample.py:92-95``, ``coco_sample.py:97-100``, ``visdrone_sample.py:84-88``).

Same signatures: ``recover_bboxes_prediction_{voc,coco,visdrone}(batch_pred_bbox, input_size, batch_original_size)`` with
``batch_pred_bbox`` (B, N, 5+C), ``input_size`` (2,) = (h, w), ``batch_original_size`` (B, 2) or (2,).  Returns (B, N, 4+C)
and, like the reference, leaves the recovered coordinates / conf-scaled probabilities in the input tensor.  The per-image
affine parameters (B x 3 scalars) are computed with a few tiny tensor ops; the per-box work runs in ``pq_recover``.
"""
from typing import Callable, Dict

import torch

from .. import ops


def _letterbox_affine(input_size: torch.Tensor, orig: torch.Tensor):
    ratio = (input_size / orig).min(dim=-1)[0]
    delta = ((input_size - (ratio.unsqueeze(-1) * orig).round()) / 2).floor()
    return delta, ratio


def _visdrone_affine(input_size: torch.Tensor, orig: torch.Tensor):
    ratio = 1.25
    real_in = torch.ceil(ratio * orig / 32) * 32
    delta = ((real_in - ratio * orig) / 2).floor()
    return delta, torch.full(orig.shape[:-1] if orig.dim() == 2 else (1,), ratio, device=orig.device)


def recover_bboxes_prediction(batch_pred_bbox, input_size, batch_original_size, affine_func):
    orig = batch_original_size.to(batch_pred_bbox.device, torch.float32)
    delta, ratio = affine_func(input_size.to(orig.device, torch.float32), orig)
    pred = batch_pred_bbox
    if not (pred.is_contiguous() and pred.dtype == torch.float32):
        raise ValueError('recover_bboxes_prediction expects a contiguous float32 prediction tensor')
    return ops.recover(pred, delta, ratio, orig, mutate=True)


def recover_bboxes_prediction_voc(batch_pred_bbox, input_size, batch_original_size):
    return recover_bboxes_prediction(batch_pred_bbox, input_size, batch_original_size, _letterbox_affine)


def recover_bboxes_prediction_coco(batch_pred_bbox, input_size, batch_original_size):
    return recover_bboxes_prediction(batch_pred_bbox, input_size, batch_original_size, _letterbox_affine)


def recover_bboxes_prediction_visdrone(batch_pred_bbox, input_size, batch_original_size):
    return recover_bboxes_prediction(batch_pred_bbox, input_size, batch_original_size, _visdrone_affine)


RECOVER_BBOXES_REGISTER: Dict[str, Callable] = {
    'voc': recover_bboxes_prediction_voc,
    'coco': recover_bboxes_prediction_coco,
    'visdrone': recover_bboxes_prediction_visdrone,
}
