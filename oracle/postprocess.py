"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the steps either side of the network:

* ``recover_boxes``     -- reference ``dataset/base_sample.py:98-139`` with the affine helpers
  ``dataset/voc_sample.py:92-95`` (= ``coco_sample.py:97-100``) and ``dataset/visdrone_sample.py:84-88``
* ``torch_nms``         -- reference ``tools.py:540-566``; the call into ``torchvision.ops.boxes.batched_nms``
  (``tools.py:556``; torchvision==0.6.0, third party, absent) is restated from torchvision's published algorithm:
  ``boxes + cls * (boxes.max() + 1)``, areas ``(x2-x1)*(y2-y1)``, descending score order, suppress iff
  ``inter / (area_i + area_j - inter) > thr``.  PARITY UNPINNED for that one call (tie order among equal scores
  is fixed here as ascending candidate index).
* ``numpy_nms``         -- reference ``tools.py:507-538`` (in-repo greedy NMS / soft-NMS; pinned by fixture G7)
* ``create_label``      -- reference ``dataset/train_dataset.py:109-150``; ``pad_boxes`` = ``_pad_arrays`` :16-24
"""
import numpy as np
import torch

from .loss import iou_xywh_np, iou_xyxy_np


# ------------------------------------------------------------------ recover
def _affine_letterbox(input_size, orig):
    """voc_sample.py:92-95: ratio = min(in/orig); delta = floor((in - round(ratio*orig)) / 2)."""
    ratio = (input_size / orig).min(dim=-1)[0].unsqueeze(-1)          # (B,1) | (1,)
    delta = ((input_size - (ratio * orig).round()) / 2).floor()        # (B,2) | (2,)
    return delta, ratio.unsqueeze(-1)                                  # the reference unsqueezes twice


def _affine_visdrone(input_size, orig):
    """visdrone_sample.py:84-88: fixed ratio 1.25, input recomputed as ceil(1.25*orig/32)*32."""
    ratio = 1.25
    real_in = torch.ceil(ratio * orig / 32) * 32
    return ((real_in - ratio * orig) / 2).floor(), ratio


AFFINE = {'voc': _affine_letterbox, 'coco': _affine_letterbox, 'visdrone': _affine_visdrone}


def recover_boxes(pred, input_size, orig_size, dataset='voc'):
    """pred (B,N,5+C) decoded, input_size (2,) = (h,w), orig_size (B,2)|(2,) = (h,w)  ->  (B,N,4+C).

    Unlike the reference this does not mutate ``pred`` (values are identical)."""
    pred = pred.clone()
    box, conf, prob = pred[..., :4], pred[..., 4:5], pred[..., 5:]
    delta, ratio = AFFINE[dataset](input_size, orig_size)
    shift = delta[..., [1, 0]].repeat(1, 2).unsqueeze(1)               # (B,1,4) | (1,1,4), order x,y,x,y
    box = (box - shift) / ratio
    edge = (orig_size - 1)[..., [1, 0]]
    if orig_size.dim() == 2:
        edge = edge.unsqueeze(1)
    lo = box[..., :2].clamp_min(0)
    hi = torch.min(box[..., 2:], edge)
    return torch.cat([lo, hi, prob * conf], -1)


# ------------------------------------------------------------------ NMS
def greedy_nms_f32(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """torchvision ``nms`` (CPU kernel) in fp32 arithmetic.  Returns kept indices, score-descending."""
    boxes = boxes.astype(np.float32)
    order = np.argsort(-scores.astype(np.float32), kind='stable')
    x1, y1, x2, y2 = (boxes[:, i] for i in range(4))
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(len(boxes), dtype=bool)
    keep = []
    thr = np.float32(thr)
    for pos, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        rest = rest[~dead[rest]]
        if rest.size == 0:
            continue
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (area[i] + area[rest] - inter)
        dead[rest[ovr > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms_f32(boxes, scores, cls, thr):
    """torchvision ``batched_nms``: per-class NMS through a coordinate offset."""
    if len(boxes) == 0:
        return np.zeros((0,), dtype=np.int64)
    boxes = boxes.astype(np.float32)
    off = cls.astype(np.float32) * (boxes.max() + np.float32(1))
    return greedy_nms_f32(boxes + off[:, None], scores, thr)


def nms_candidates(bboxes: np.ndarray, score_thr: float):
    """tools.py:550-555: multi-label candidates in row-major (box, class) order."""
    sc = bboxes[:, 4:]
    rows, cols = np.nonzero(sc > np.float32(score_thr))
    return bboxes[rows, :4], sc[rows, cols], cols.astype(np.int64), rows.astype(np.int64)


def torch_nms(bboxes: np.ndarray, score_thr: float, iou_thr: float, return_index=False):
    """tools.py:540-566 on numpy fp32 input (N,4+C) -> (K,6) or shape-(0,) when nothing survives."""
    bboxes = np.asarray(bboxes, dtype=np.float32)
    pb, ps, pc, rows = nms_candidates(bboxes, score_thr)
    keep = batched_nms_f32(pb, ps, pc, iou_thr)
    if keep.size == 0:
        out = np.zeros((0,), dtype=np.float32)
        return (out, keep, rows, pc) if return_index else out
    out = np.concatenate([pb[keep], ps[keep, None], pc[keep, None].astype(np.float32)], 1)
    return (out, keep, rows, pc) if return_index else out


def numpy_nms(bboxes, score_threshold, iou_threshold, sigma=0.3, method='nms'):
    """tools.py:507-538.  bboxes (N,6) = x1,y1,x2,y2,score,class (already arg-maxed)."""
    assert method in ('nms', 'soft-nms')
    out = []
    for c in list(set(bboxes[:, 5])):
        pool = bboxes[bboxes[:, 5] == c]
        while len(pool) > 0:
            j = int(np.argmax(pool[:, 4]))
            top = pool[j]
            out.append(top)
            pool = np.concatenate([pool[:j], pool[j + 1:]])
            iou = iou_xyxy_np(top[np.newaxis, :4], pool[:, :4])
            if method == 'nms':
                wgt = np.ones((len(iou),), dtype=np.float32)
                wgt[iou > iou_threshold] = 0.0
            else:
                wgt = np.exp(-(1.0 * iou ** 2 / sigma))
            pool[:, 4] = pool[:, 4] * wgt
            pool = pool[pool[:, 4] > score_threshold]
    return np.array(out)


# ------------------------------------------------------------------ label assignment
DEFAULT_ANCHORS = np.array([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119),
                            (116, 90), (156, 198), (373, 326)], dtype=np.float32)   # config.py:58-59
DEFAULT_STRIDES = np.array([8, 16, 32])                                              # config.py:56


def create_label(bboxes, output_sizes, num_classes, strides=DEFAULT_STRIDES, anchors=DEFAULT_ANCHORS,
                 gt_per_grid=3, iou_threshold=0.3):
    """train_dataset.py:109-150.  bboxes: iterable of [x1,y1,x2,y2,cls,mixw]; output_sizes (3,2) = (h,w)."""
    label = [np.zeros((output_sizes[i][0], output_sizes[i][1], gt_per_grid, 6 + num_classes), np.float32)
             for i in range(3)]
    for l in label:
        l[..., -1] = 1.0
    per_scale = [[] for _ in range(3)]
    for bb in bboxes:
        coor = bb[:4]
        cls, mixw = int(bb[4]), bb[5]
        xywh = np.concatenate([(coor[2:] + coor[:2]) * 0.5, coor[2:] - coor[:2]], -1)
        onehot = np.zeros(num_classes, dtype=np.float64)
        onehot[cls] = 1.0
        smooth = onehot * (1 - 0.01) + 0.01 * np.full(num_classes, 1.0 / num_classes)
        cell = (xywh[:2][:, None] // strides).astype(np.int32).T                 # (3,2) = (x,y) per scale
        centre = (cell.astype(np.float32) + 0.5) * strides[:, None]
        cand = np.concatenate([np.repeat(centre, 3, axis=0), anchors], -1)       # (9,4) xywh
        ious = iou_xywh_np(xywh, cand)
        hit = ious > iou_threshold
        if not hit.any():
            hit[ious.argmax()] = 1
        for k in hit.nonzero()[0]:
            s, a = k // 3, k % 3
            x, y = cell[s]
            label[s][y, x, a, :] = np.concatenate([coor, [1.0], smooth, [mixw]], -1)
            per_scale[s].append(coor)
    return (*label, *per_scale)


def pad_boxes(lists):
    """train_dataset.py:16-24: pad each image's box list with zero rows to the batch max (>=1)."""
    n = max(max(len(a) for a in lists), 1)
    out = []
    for a in lists:
        if len(a) == 0:
            out.append(np.zeros((n, 4), np.float32))
        else:
            out.append(np.pad(np.asarray(a, dtype=np.float32), ((0, n - len(a)), (0, 0)), 'constant'))
    return np.stack(out, 0)


def synth_targets(batch, size, num_classes, seed=1, max_boxes=8):
    """Synthetic training targets: per image n~U{1..max_boxes} boxes, w,h in [16, size/2]."""
    rng = np.random.RandomState(seed)
    hw = np.array([[size // s, size // s] for s in DEFAULT_STRIDES])
    labels, boxes = [[], [], []], [[], [], []]
    for _ in range(batch):
        n = rng.randint(1, max_boxes + 1)
        bbs = []
        for _ in range(n):
            w, h = rng.uniform(16, size * 0.5, 2)
            x1, y1 = rng.uniform(0, size - w), rng.uniform(0, size - h)
            bbs.append(np.array([x1, y1, x1 + w, y1 + h, rng.randint(0, num_classes), 1.0], np.float32))
        out = create_label(bbs, hw, num_classes)
        for i in range(3):
            labels[i].append(out[i])
            boxes[i].append(out[3 + i])
    return tuple(torch.from_numpy(np.stack(l, 0)) for l in labels) + \
        tuple(torch.from_numpy(pad_boxes(b)) for b in boxes)
