"""Label assignment on the host (numpy) -- mirrors the reference ``TrainDataset.create_label`` and the box padding of
``collate_batch`` (``dataset/train_dataset.py:16-43,109-150``; anchors / strides / IoU threshold defaults from
``config.py:56-61``).  This is the producer of the 6-tuple ``DetectionModel.forward(x, target)`` consumes; it runs in
DataLoader workers in the reference; :func:`create_label_batch` is the device version for a whole batch (``pq_create_label``),
bit-identical to the host function.
"""
from typing import List, Sequence, Tuple

import numpy as np
import torch

from ..tools import iou_xywh_numpy

ANCHORS = np.array([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)],
                   dtype=np.float32)
STRIDES = np.array([8, 16, 32])


def create_label(bboxes: Sequence[np.ndarray], output_sizes: np.ndarray, num_classes: int, strides=STRIDES,
                 anchors=ANCHORS, gt_per_grid: int = 3, anchors_iou_threshold: float = 0.3):
    """bboxes rows = [x1, y1, x2, y2, class, mixup weight]; output_sizes (3, 2) = (h, w) of the three heads.
    -> (label_s, label_m, label_l, boxes_s, boxes_m, boxes_l); labels are (h, w, gt_per_grid, 6 + C) float32.

    One pass over ALL boxes of the image: cells, the 9 anchor IoUs and the threshold / arg-max choice are computed as
    (n, 9) arrays; only the final writes run in (box, anchor) order, because a later ground truth overwrites an earlier one
    in the same cell slot (reference semantics, pinned by fixture G8 and by the device kernel ``pq_create_label``).  The
    integer / float32 / float64 promotions of the reference's NumPy expressions are kept (cells from float32 // int64,
    cell centres in float64 through the int64 strides), so the grids are bit-identical."""
    grids = [np.zeros((output_sizes[i][0], output_sizes[i][1], gt_per_grid, 6 + num_classes), dtype=np.float32)
             for i in range(3)]
    for gmap in grids:
        gmap[..., -1] = 1.0                                       # mix-up weight defaults to 1
    lists: List[List[np.ndarray]] = [[], [], []]
    boxes = np.asarray(bboxes)                 # the boxes' own dtype decides the arithmetic, as in the reference's per-box loop:
    if boxes.dtype not in (np.float32, np.float64):   # float64 boxes take centres, // strides and the anchor IoU in float64
        boxes = boxes.astype(np.float32)
    boxes = boxes.reshape(-1, 6)
    n = boxes.shape[0]
    if n == 0:
        return (*grids, *lists)
    strides = np.asarray(strides)
    corners = boxes[:, :4]
    centre_size = np.concatenate([(corners[:, 2:] + corners[:, :2]) * 0.5, corners[:, 2:] - corners[:, :2]], axis=-1)   # (n, 4) f32
    cells = (centre_size[:, :2, None] // strides).astype(np.int32).transpose(0, 2, 1)                 # (n, 3, 2): (x, y) per scale
    cell_centres = (cells.astype(np.float32) + 0.5) * strides[None, :, None]                         # float64 via int64 strides
    anchor_boxes = np.concatenate([np.repeat(cell_centres, gt_per_grid, axis=1),
                                   np.broadcast_to(anchors, (n,) + anchors.shape)], axis=-1)          # (n, 9, 4)
    overlap = iou_xywh_numpy(centre_size[:, None, :], anchor_boxes)                                   # (n, 9)
    chosen = overlap > anchors_iou_threshold
    none = ~chosen.any(axis=1)
    chosen[none, overlap[none].argmax(axis=1)] = True                                                 # fallback: the best anchor
    # rows to write: [corners, 1, smoothed one-hot (float64 arithmetic, stored as float32), mix]
    prob = np.zeros((n, num_classes), dtype=np.float64)
    prob[np.arange(n), boxes[:, 4].astype(np.int64)] = 1.0
    prob = prob * (1 - 0.01) + 0.01 * np.full(num_classes, 1.0 / num_classes)                         # label smoothing, delta = 0.01
    rows = np.concatenate([corners, np.ones((n, 1)), prob, boxes[:, 5:6]], axis=-1).astype(np.float32)
    for b, k in zip(*np.nonzero(chosen)):                                                             # box-major, anchor-minor order
        scale, slot = k // gt_per_grid, k % gt_per_grid
        cx, cy = cells[b, scale]
        grids[scale][cy, cx, slot, :] = rows[b]
        lists[scale].append(corners[b])
    return (*grids, *lists)


def pad_box_lists(per_image: Sequence[Sequence[np.ndarray]]) -> np.ndarray:
    """zero-pad every image's list to the batch maximum (at least one row) -> (B, M, 4) float32."""
    m = max(max(len(a) for a in per_image), 1)
    out = np.zeros((len(per_image), m, 4), dtype=np.float32)
    for i, a in enumerate(per_image):
        if len(a):
            out[i, :len(a)] = np.asarray(a, dtype=np.float32)
    return out


def synthetic_targets(batch: int, size: int, num_classes: int, seed: int = 1, max_boxes: int = 8) -> Tuple[torch.Tensor, ...]:
    """SURVEY 8(d) synthetic targets: per image n ~ U{1..max_boxes} boxes with w, h in [16, size/2], uniform classes."""
    rng = np.random.RandomState(seed)
    hw = np.array([[size // s, size // s] for s in STRIDES])
    grids, lists = [[], [], []], [[], [], []]
    for _ in range(batch):
        n = rng.randint(1, max_boxes + 1)
        boxes = []
        for _ in range(n):
            w, h = rng.uniform(16, size * 0.5, 2)
            x1, y1 = rng.uniform(0, size - w), rng.uniform(0, size - h)
            boxes.append(np.array([x1, y1, x1 + w, y1 + h, rng.randint(0, num_classes), 1.0], dtype=np.float32))
        out = create_label(boxes, hw, num_classes)
        for i in range(3):
            grids[i].append(out[i])
            lists[i].append(out[3 + i])
    return tuple(torch.from_numpy(np.stack(g, 0)) for g in grids) + tuple(torch.from_numpy(pad_box_lists(l)) for l in lists)


def create_label_batch(gt: torch.Tensor, counts, size_hw, num_classes: int, trim: bool = True):
    """Device label assignment for a batch: ``gt`` (B, G, 6) on the GPU (rows past ``counts[b]`` ignored), input size
    ``(H, W)`` -> the 6-tuple ``DetectionModel.forward`` takes.  ``trim=True`` cuts the box lists to the batch maximum like
    the reference's ``collate_batch`` (one host synchronisation); ``trim=False`` keeps the 3*G capacity -- equivalent for the
    loss, which ignores all-zero rows -- and never synchronises."""
    from .. import ops
    h, w = size_hw
    out_hw = [(h // s, w // s) for s in STRIDES]
    ls, lm, ll, bs, bm, bl, cnt = ops.create_label(gt, counts, out_hw, num_classes)
    if trim:
        m = cnt.max(0).values.clamp(min=1).tolist()
        bs, bm, bl = bs[:, :m[0]].contiguous(), bm[:, :m[1]].contiguous(), bl[:, :m[2]].contiguous()
    return ls, lm, ll, bs, bm, bl
