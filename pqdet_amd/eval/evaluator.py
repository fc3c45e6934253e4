"""Evaluation harness around the device post-process: model -> recover -> NMS -> mAP (reference ``eval/evaluator.py``).

What changes against the reference's loop (``evaluator.py:44-62``): the reference runs ``torch_nms`` image by image and
copies every result to the host before starting the next image (one device synchronisation per image).  Here all images
of a batch are queued on the stream back to back (``pq_recover`` once, ``pq_nms`` per image into one output slab) and the
host reads the per-image counts and rows once per batch.  The per-class matching / precision-recall bookkeeping is host
work in the reference as well (NumPy); it is restated here in vectorised form.

``Evaluator(model, loader, dataset_name=..., classes=..., input_size=..., score_threshold=..., iou_threshold=...)``
takes the settings the reference reads from its yacs config (``config.eval.*``, ``config.dataset.*``); a config object
with those attributes can be passed as ``config=`` instead.
"""
from collections import defaultdict, namedtuple
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from ..dataset import RECOVER_BBOXES_REGISTER

AP_IOU_THRESHOLDS = np.linspace(0.5, 0.95, 10)

AP = namedtuple('AP', ['mAPs', 'APs', 'AP', 'raw', 'class_names', 'iou_thresholds'])      # reference tools.py:37


def average_precision(recs: np.ndarray, precs: np.ndarray) -> np.ndarray:
    """Area under the monotone precision envelope, one value per IoU threshold row (``evaluator.py:143-157``)."""
    mrec = np.pad(recs, ((0, 0), (1, 1)), constant_values=(0., 1.))
    mpre = np.pad(precs, ((0, 0), (1, 1)), constant_values=0.)
    # precision envelope: running maximum taken from the right
    env = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]
    return np.sum(np.diff(mrec) * env[:, 1:], axis=1)


class _Gt:
    __slots__ = ('boxes', 'seen', 'difficult')

    def __init__(self, boxes, difficult):
        self.boxes, self.difficult = boxes, difficult
        self.seen = np.zeros((len(AP_IOU_THRESHOLDS), len(boxes)), dtype=bool)


class Evaluator:
    def __init__(self, model: Callable[[torch.Tensor], torch.Tensor], dataset: Optional[Iterable] = None, config=None, *,
                 dataset_name: str = 'voc', classes: Sequence[str] = (), input_size=(512, 512),
                 score_threshold: float = 0.1, iou_threshold: float = 0.45):
        if config is not None:
            score_threshold, iou_threshold = config.eval.score_threshold, config.eval.iou_threshold
            input_size = config.eval.input_size
            dataset_name, classes = config.dataset.name, config.dataset.classes
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        self._score_threshold, self._iou_threshold = float(score_threshold), float(iou_threshold)
        self._input_size = tuple(input_size)
        self._recover_bboxes = RECOVER_BBOXES_REGISTER[dataset_name.lower()]
        self._classes = list(classes)
        self.model, self.dataset = model, dataset
        self.init_statics()

    def init_statics(self):
        self.detections_count = 0
        self._dets: Dict[int, List] = defaultdict(list)        # class -> [(file, row, insertion index)]
        self.labels: Dict[str, Dict[int, _Gt]] = defaultdict(dict)
        self.gt_count: Dict[int, int] = defaultdict(int)
        self._ticket = 0

    # ---- device side -----------------------------------------------------------------------------------
    def predict(self, imgs: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.model(imgs)

    def detect(self, imgs: torch.Tensor, shapes: torch.Tensor) -> List[np.ndarray]:
        """Batch of input images + their original (h, w) -> per-image (K, 6) arrays x1,y1,x2,y2,score,class in
        score-descending order (what ``torch_nms(...).cpu().numpy()`` returns per image, empty -> shape (0,))."""
        pred = self.predict(imgs)
        dev = pred.device
        boxes = self._recover_bboxes(pred, torch.FloatTensor(self._input_size).to(dev), shapes.to(dev))
        b, n, w = boxes.shape
        cap = n * (w - 4)
        if b > 1 and b * (w - 4) < 65535:
            # the whole batch in one pq_nms call (images separated like one torch_nms call each); ONE device -> host copy
            dets, img, cnt = ops.nms_batch(boxes, self._score_threshold, self._iou_threshold, cap=self._batch_cap(b, n, w - 4))
            k = int(cnt.item())
            if 0 <= k <= dets.shape[0]:
                order = torch.sort(img[:k], stable=True).indices          # per-image lists, score order kept
                rows = torch.cat([dets[:k][order], img[:k][order].float().unsqueeze(1)], 1).cpu().numpy()
                res = []
                for i in range(b):
                    mine = rows[rows[:, 6] == i][:, :6]
                    res.append(mine if len(mine) else np.zeros((0,), dtype=np.float32))
                return res
            ops.nms_grow(n, w - 4)              # candidate / output capacity exceeded: per-image path below sorts everything
        outs = [ops.nms(boxes[i], self._score_threshold, self._iou_threshold, cap=cap) for i in range(b)]
        counts = torch.cat([c for _, c in outs]).cpu().tolist()                     # the one synchronisation per batch
        if any(k < 0 for k in counts):          # candidate capacity exceeded somewhere: redo those images over all pairs
            ops.nms_grow(n, w - 4)
            for i, k in enumerate(counts):
                if k < 0:
                    outs[i] = ops.nms(boxes[i], self._score_threshold, self._iou_threshold, cap=cap, cand_cap=0)
                    counts[i] = int(outs[i][1].item())
        res = []
        for (dets, _), k in zip(outs, counts):
            res.append(dets[:k].cpu().numpy() if k else np.zeros((0,), dtype=np.float32))
        return res

    @staticmethod
    def _batch_cap(b: int, n: int, ncls: int) -> int:
        """rows the batched NMS may return: every candidate the adaptive sort capacity admits can survive"""
        return max(1024, min(b * n * ncls, b * ops.nms_capacity(n, ncls)))

    def evaluate(self) -> AP:
        for batch_img, batch_file_name, batch_img_shape, batch_label, batch_diff in self.dataset:
            dets = self.detect(batch_img, batch_img_shape)
            for file_name, labels, diffs, bboxes in zip(batch_file_name, batch_label, batch_diff, dets):
                self.add_detections(file_name, bboxes)
                self.add_labels(file_name, np.asarray(labels), np.asarray(diffs))
        return self.AP()

    # ---- host bookkeeping -------------------------------------------------------------------------------
    def add_detections(self, file_name: str, bboxes: np.ndarray):
        self.detections_count += len(bboxes)
        for row in bboxes:
            self._dets[int(row[-1])].append((file_name, row, self._ticket))
            self._ticket += 1

    def add_labels(self, file_name: str, bboxes: np.ndarray, diffs: np.ndarray):
        cls = bboxes[:, -1].astype(int)
        for c in set(cls.tolist()):
            sel = cls == c
            d = diffs[sel].astype(bool)
            order = np.argsort(d)                       # non-difficult boxes first
            self.labels[file_name][c] = _Gt(bboxes[sel][:, :4][order], d[order])
            self.gt_count[c] += int(np.sum(~d))

    def _match(self, gt: _Gt, box: np.ndarray, tp: np.ndarray, fp: np.ndarray, col: int):
        b = gt.boxes
        iw = np.maximum(np.minimum(b[:, 2], box[2]) - np.maximum(b[:, 0], box[0]) + 1., 0.)
        ih = np.maximum(np.minimum(b[:, 3], box[3]) - np.maximum(b[:, 1], box[1]) + 1., 0.)
        inter = iw * ih
        union = (box[2] - box[0] + 1.) * (box[3] - box[1] + 1.) + (b[:, 2] - b[:, 0] + 1.) * (b[:, 3] - b[:, 1] + 1.) - inter
        ov = inter / union
        for t, thr in enumerate(AP_IOU_THRESHOLDS):
            pick, best = -1, min(thr, 1 - 1e-10)
            for j in range(len(ov)):
                if gt.seen[t, j]:
                    continue
                if pick > -1 and not gt.difficult[pick] and gt.difficult[j]:
                    break                               # difficult boxes are sorted last: stop once a normal match exists
                if ov[j] < best:
                    continue
                pick, best = j, ov[j]
            if gt.difficult[pick]:                      # pick == -1 reads the last box, like the reference's index -1
                continue
            if pick == -1:
                fp[t, col] = 1
            else:
                tp[t, col] = 1
                gt.seen[t, pick] = True

    def AP(self) -> AP:
        nthr = len(AP_IOU_THRESHOLDS)
        raw = np.zeros((len(self._classes), nthr))
        metrics = None
        for c, dets in self._dets.items():
            dets = sorted(dets, key=lambda d: (-d[1][4], d[2]))        # score descending, earlier insertion first
            tp, fp = np.zeros((nthr, len(dets))), np.zeros((nthr, len(dets)))
            for col, (file_name, row, _) in enumerate(dets):
                gt = self.labels[file_name].get(c)
                if gt is None:
                    fp[:, col] = 1
                else:
                    self._match(gt, row[:4], tp, fp, col)
            tp, fp = np.cumsum(tp, axis=1), np.cumsum(fp, axis=1)
            with np.errstate(invalid='ignore', divide='ignore'):
                rec = tp / self.gt_count[c]                            # 0/0 -> NaN for a class that was never labelled
            prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
            raw[c] = average_precision(rec, prec)
            maps = np.mean(raw, axis=0)
            metrics = AP(maps, np.mean(raw, axis=1), np.mean(maps), raw, self._classes, AP_IOU_THRESHOLDS)
        self.init_statics()
        return metrics
