"""Host-side mirror of the hot-path functions of the reference ``tools.py``.

Kept names / signatures (reference ``tools.py`` line numbers in brackets): ``build_model`` [158-243], ``load_weight``
[120-136], ``load_backbone`` [138-156], ``bare_model`` [251-254], ``get_bn_layers`` [316-333], ``iou_calc3`` [357-376],
``giou`` [378-404], ``diou`` [406-437], ``ciou`` [439-477], ``iou_calc1`` [335-355], ``iou_xywh_numpy`` [479-505],
``torch_nms`` [540-566], ``compute_time`` [74-94].

The GPU arithmetic is in libpqdet_hip.so; numpy helpers that the reference runs on the host stay numpy.
Data-parallel training is one process per GPU (:mod:`pqdet_amd.parallel`), so ``dataparallel=True`` does not wrap the model
in ``nn.DataParallel``: with an initialised ``torch.distributed`` process group it returns a
:class:`pqdet_amd.parallel.DistributedModel`, otherwise the bare model.
"""
from io import StringIO
from time import time_ns
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import abi, ops


# ----------------------------------------------------------------------------------------------- IoU family
def _pairs(kind):
    def fn(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
        return ops.iou_pairs(boxes1, boxes2, kind)
    return fn


iou_calc3 = _pairs(abi.BOX_IOU)
giou = _pairs(abi.BOX_GIOU)
diou = _pairs(abi.BOX_DIOU)
ciou = _pairs(abi.BOX_CIOU)


def iou_calc1(boxes1: np.ndarray, boxes2: np.ndarray):
    """xyxy IoU on the host, union clamped at 1e-14 (reference :335-355)."""
    a1 = (boxes1[..., 2] - boxes1[..., 0]) * (boxes1[..., 3] - boxes1[..., 1])
    a2 = (boxes2[..., 2] - boxes2[..., 0]) * (boxes2[..., 3] - boxes2[..., 1])
    lo = np.maximum(boxes1[..., :2], boxes2[..., :2])
    hi = np.minimum(boxes1[..., 2:], boxes2[..., 2:])
    wh = np.maximum(hi - lo, 0.0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / np.maximum(a1 + a2 - inter, 1e-14)


def iou_xywh_numpy(boxes1: np.ndarray, boxes2: np.ndarray):
    """centre-size IoU on the host (reference :479-505); used by label assignment."""
    a1 = boxes1[..., 2] * boxes1[..., 3]
    a2 = boxes2[..., 2] * boxes2[..., 3]
    c1 = np.concatenate([boxes1[..., :2] - boxes1[..., 2:] * 0.5, boxes1[..., :2] + boxes1[..., 2:] * 0.5], axis=-1)
    c2 = np.concatenate([boxes2[..., :2] - boxes2[..., 2:] * 0.5, boxes2[..., :2] + boxes2[..., 2:] * 0.5], axis=-1)
    wh = np.maximum(np.minimum(c1[..., 2:], c2[..., 2:]) - np.maximum(c1[..., :2], c2[..., :2]), 0.0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (a1 + a2 - inter)


# ----------------------------------------------------------------------------------------------- NMS
def torch_nms(bboxes: torch.Tensor, score_threshold: float, iou_threshold: float) -> torch.Tensor:
    """reference :540-566.  bboxes (N, 4+C) recovered predictions -> (K, 6) = x1,y1,x2,y2,score,class on the input
    device in score-descending order, or a shape-(0,) tensor when nothing survives.

    The whole pipeline (candidate marking, sort, per-class greedy suppression, compaction) runs on the GPU inside
    ``pq_nms``; the only host synchronisation is reading K to size the returned view (the reference synchronises at the
    same point through ``nonzero()`` / ``.cpu()``)."""
    if bboxes.numel() == 0:
        return torch.tensor([]).to(bboxes)
    dets, count = ops.nms(bboxes, float(score_threshold), float(iou_threshold))
    k = int(count.item())
    if k < 0:          # more candidates than the compacting fast path was sized for: sort all N*C pairs instead
        ops.nms_grow(bboxes.shape[0], bboxes.shape[1] - 4)
        dets, count = ops.nms(bboxes, float(score_threshold), float(iou_threshold), cand_cap=0)
        k = int(count.item())
    if k == 0:
        return torch.tensor([]).to(bboxes)
    return dets[:k]


def nms(bboxes: np.ndarray, score_threshold: float, iou_threshold: float, sigma: float = 0.3, method: str = 'nms'):
    """Host-side per-class NMS on arg-maxed detections (reference :507-538; not on the evaluator path, kept for API parity).

    ``bboxes`` (N, 6) = x1,y1,x2,y2,score,class.  Per class (in ``list(set(classes))`` order): take the best remaining
    box, then either drop what overlaps it by more than ``iou_threshold`` (``'nms'``) or decay the remaining scores by
    ``exp(-iou^2/sigma)`` (``'soft-nms'``, the input rows' scores are modified like in the reference), and drop scores
    ``<= score_threshold``.  Hard NMS is done from one score-sorted pass over a precomputed IoU matrix."""
    assert method in ['nms', 'soft-nms']
    picked = []
    for cls in list(set(bboxes[:, 5])):
        rows = bboxes[bboxes[:, 5] == cls]
        n = len(rows)
        if method == 'nms':
            order = np.argsort(-rows[:, 4], kind='stable')          # equal scores: first occurrence first, like argmax
            cand = rows[order]
            alive = cand[:, 4] > score_threshold
            alive[0] = True                                          # the best box of a class is always reported
            iou = iou_calc1(cand[:, None, :4], cand[None, :, :4])
            for i in range(n):
                if alive[i]:
                    picked.append(cand[i])
                    alive[i + 1:] &= ~(iou[i, i + 1:] > iou_threshold)
        else:
            score = rows[:, 4].copy()
            alive = np.ones(n, dtype=bool)
            while alive.any():
                j = int(np.argmax(np.where(alive, score, -np.inf)))
                best = rows[j].copy()
                best[4] = score[j]
                picked.append(best)
                alive[j] = False
                idx = np.nonzero(alive)[0]
                w = np.exp(-(1.0 * iou_calc1(rows[j][np.newaxis, :4], rows[idx, :4]) ** 2 / sigma))
                score[idx] = (score[idx] * w).astype(rows.dtype)
                alive[idx] = score[idx] > score_threshold
    return np.array(picked)


# ----------------------------------------------------------------------------------------------- weights / build
def _state_dict_is_dp(state_dict: Dict) -> bool:
    for k in state_dict:
        return k.startswith('module.')
    return False


def _model_has_module_prefix(model: nn.Module) -> bool:
    return hasattr(model, 'module') and isinstance(getattr(model, 'module'), nn.Module)


def bare_model(model: nn.Module) -> nn.Module:
    return model.module if _model_has_module_prefix(model) else model


def _strip(state_dict: Dict) -> Dict:
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}


def load_weight(model: nn.Module, state_dict: Dict):
    """Keys must match exactly, up to an optional ``module.`` prefix on either side (reference :120-136)."""
    bare_model(model).load_state_dict(_strip(state_dict))


def load_backbone(model: nn.Module, state_dict: Dict):
    """``state_dict`` holds a prefix of the model's layers (reference :138-156)."""
    target = bare_model(model)
    full = target.state_dict()
    full.update(_strip(state_dict))
    target.load_state_dict(full)


def build_model(cfg_path: Optional[str], weight_path: Optional[str] = None, backbone_path: Optional[str] = None,
                clear_history: bool = False, device='cuda', dataparallel: bool = True, device_ids=None,
                qat: bool = False, backend: Optional[str] = None, quantized: bool = False, onnx: bool = False):
    """reference :158-243 -> (model, info).  ``info`` = checkpoint dict minus ``'model'`` (empty without a weight file)."""
    from .model.interpreter import DetectionModel
    model_info = {}
    state_dict_type = None
    if weight_path:
        ckpt = torch.load(weight_path, map_location='cpu', weights_only=False)      # the file holds the trainer's AP record too
        state_dict_type = ckpt.get('type', 'normal')
        weight = ckpt['model']
        model_info = {k: v for k, v in ckpt.items() if k != 'model'}
        if clear_history:
            model_info['step'] = 0
    cfg = cfg_path if cfg_path else StringIO(ckpt['cfg'])
    if state_dict_type == 'quant' or quantized:
        raise NotImplementedError('converted int8 models (torch.quantization.convert, CPU-only in the reference) are not built')
    qat = qat or state_dict_type == 'qat'
    if qat:                                   # reference :229-232: the recipe comes from the caller, else from the checkpoint
        backend = backend or model_info.get('backend')
        if backend in (None, 'none'):
            backend = 'qnnpack'
        if backend != 'qnnpack':
            raise NotImplementedError("QAT backend %r: only the 'qnnpack' recipe (yamls/quant.yaml) is implemented" % (backend,))
    model = DetectionModel(cfg, qat, onnx)
    if backbone_path:
        print('loading backbone weights from {}'.format(backbone_path))
        load_backbone(model, torch.load(backbone_path, map_location='cpu', weights_only=False))
    if state_dict_type == 'normal':
        load_weight(model, weight)
        print('resumed at %d steps from %s' % (model_info['step'], weight_path))
    model = model.to(device)
    if qat:                                   # reference :225-235: fuse conv+bn+relu, then attach observers / fake-quantisers
        model.train()
        fuse_model(model)
        prepare_qat(model, backend)
        if state_dict_type == 'qat':          # reference :236-238: a QAT checkpoint is loaded into the prepared model
            load_qat_state_dict(model, weight)
        return model, model_info              # QAT is single-GPU in the reference (trainer.py:148)
    if dataparallel:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from .parallel import DistributedModel
            model = DistributedModel(model)
    return model, model_info


def fuse_model(model: nn.Module, inplace: bool = True) -> nn.Module:
    """reference :256-275.  Conv + BatchNorm + ReLU are always executed fused by the HIP engine; kept for API parity."""
    return model


def prepare_qat(model: nn.Module, backend: str = 'qnnpack', inplace: bool = True) -> nn.Module:
    """reference :277-290: per-tensor fake-quantisation of every conv weight (qint8, symmetric) and of the input, every conv
    block output, shortcut sum and concatenation (quint8, affine) with moving-average min/max observers -- the
    ``get_default_qat_qconfig('qnnpack')`` recipe ('fbgemm' differs only in reduce_range / per-channel weights and is not
    offered).  The model must have been built with ``quant=True``."""
    if backend != 'qnnpack':
        raise NotImplementedError("only the 'qnnpack' QAT recipe (the reference's yamls/quant.yaml setting) is implemented")
    m = bare_model(model)
    if not getattr(m, 'quant', False):
        raise ValueError('prepare_qat needs a model built with quant=True (ReLU activations)')
    m.prepare_qat()
    return model


def disable_observer(model: nn.Module):
    """``model.apply(torch.quantization.disable_observer)`` of the reference trainer (:299-300): freeze quantiser ranges."""
    bare_model(model).qat_observer_enabled = False


_FQ_EPS = 1.1920928955078125e-07          # torch.finfo(torch.float32).eps, the observers' `eps` buffer


def qat_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """State of a prepared QAT model under the reference's key names (what ``trainer.py:194-201`` stores with
    ``'type': 'qat'``): fused modules keep BatchNorm inside ``conv`` (``module_list.i.conv.bn.weight``), every observer
    contributes ``fake_quant_enabled, observer_enabled, scale, zero_point, activation_post_process.{eps,min_val,max_val}``."""
    m = bare_model(model)
    if not getattr(m, '_qat', False):
        raise ValueError('not a prepared QAT model')
    st = m._qat_state.detach().cpu()
    on = torch.tensor([1 if m.qat_observer_enabled else 0], dtype=torch.int64)
    fq = torch.tensor([1 if m.qat_fake_quant_enabled else 0], dtype=torch.int64)

    def observer(prefix, row):
        mn, mx, sc, zp = st[row].tolist()
        return {prefix + '.fake_quant_enabled': fq.clone(), prefix + '.observer_enabled': on.clone(),
                prefix + '.scale': torch.tensor([sc], dtype=torch.float32),
                prefix + '.zero_point': torch.tensor([int(zp)], dtype=torch.int32),
                prefix + '.activation_post_process.eps': torch.tensor([_FQ_EPS], dtype=torch.float32),
                prefix + '.activation_post_process.min_val': torch.tensor(mn, dtype=torch.float32),
                prefix + '.activation_post_process.max_val': torch.tensor(mx, dtype=torch.float32)}

    out = dict(observer('qstub.activation_post_process', m._qat_index[('in',)]))
    for i, layer in enumerate(m.module_list):
        base = 'module_list.%d.' % i
        if layer._type == 'convolutional':
            out[base + 'conv.weight'] = layer.conv.weight.detach().cpu().clone()
            if layer.conv.bias is not None:
                out[base + 'conv.bias'] = layer.conv.bias.detach().cpu().clone()
            if hasattr(layer, 'bn'):
                for k, v in layer.bn.state_dict().items():
                    out[base + 'conv.bn.' + k] = v.detach().cpu().clone()
            out.update(observer(base + 'conv.weight_fake_quant', m._qat_index[('w', i)]))
            out.update(observer(base + 'conv.activation_post_process', m._qat_index[('a', i)]))
        elif ('a', i) in m._qat_index:
            out.update(observer(base + 'ffunc.activation_post_process', m._qat_index[('a', i)]))
    return out


def load_qat_state_dict(model: nn.Module, state_dict: Dict[str, torch.Tensor]):
    """Inverse of :func:`qat_state_dict`: accepts a reference QAT checkpoint's ``'model'`` entry (optional ``module.`` prefix)."""
    m = bare_model(model)
    if not getattr(m, '_qat', False):
        raise ValueError('call tools.prepare_qat(model) before loading a QAT checkpoint')
    sd = _strip(state_dict) if _state_dict_is_dp(state_dict) else state_dict
    if '_qat_state' in sd:                    # this package's own layout (``plain_state_dict``): bn.* + the observer table
        nn.Module.load_state_dict(m, sd)
        m._plans.clear()
        return
    plain = {}
    for k, v in sd.items():
        if '.conv.bn.' in k:
            plain[k.replace('.conv.bn.', '.bn.')] = v
        elif k.endswith('conv.weight') or k.endswith('conv.bias'):
            plain[k] = v
    missing = [k for k in m.plain_state_dict() if k != '_qat_state' and k not in plain]
    if missing:
        raise KeyError('QAT checkpoint lacks %d tensors, e.g. %s' % (len(missing), missing[:3]))
    with torch.no_grad():
        own = m.plain_state_dict()
        for k, v in plain.items():
            own[k].copy_(v)
        st = m._qat_state

        def take(prefix, row):
            st[row, 0] = float(sd[prefix + '.activation_post_process.min_val'])
            st[row, 1] = float(sd[prefix + '.activation_post_process.max_val'])
            st[row, 2] = float(sd[prefix + '.scale'])
            st[row, 3] = float(sd[prefix + '.zero_point'])
        take('qstub.activation_post_process', m._qat_index[('in',)])
        for key, row in m._qat_index.items():
            if key[0] == 'w':
                take('module_list.%d.conv.weight_fake_quant' % key[1], row)
            elif key[0] == 'a':
                kind = 'conv' if m.module_list[key[1]]._type == 'convolutional' else 'ffunc'
                take('module_list.%d.%s.activation_post_process' % (key[1], kind), row)
        first = 'qstub.activation_post_process'
        m.qat_observer_enabled = bool(int(sd[first + '.observer_enabled']))
        m.qat_fake_quant_enabled = bool(int(sd[first + '.fake_quant_enabled']))
    m._plans.clear()


def freeze_bn_stats(model: nn.Module):
    """``model.apply(torch.nn.intrinsic.qat.freeze_bn_stats)`` of the reference trainer (:302-303): from here on BatchNorm
    normalises with its running statistics and stops updating them, while the weights keep training."""
    bare_model(model).bn_frozen = True


def update_bn_stats(model: nn.Module):
    bare_model(model).bn_frozen = False


def enable_observer(model: nn.Module):
    bare_model(model).qat_observer_enabled = True


def get_bn_layers(model: nn.Module) -> List[nn.BatchNorm2d]:
    """BN layers eligible for the sparsity penalty (reference :316-333)."""
    out, total = [], 0
    for l in bare_model(model).module_list:
        if l._type == 'convolutional' and hasattr(l, 'bn'):
            if not hasattr(l, '_notprune'):
                out.append(l.bn)
            total += 1
    print("sparse mode: {}/{} BN layers will be sparsed.".format(len(out), total))
    return out


# ----------------------------------------------------------------------------------------------- timing helpers
class TicToc:
    """reference :595-641 (wall-clock records in ns)."""

    def __init__(self, name: Optional[str] = None):
        self.name = name
        self.last = 0
        self.records = []

    def reset(self):
        self.last = 0
        self.records.clear()

    def tic(self):
        self.last = time_ns()

    def toc(self):
        self.records.append(time_ns() - self.last)

    def mean(self):
        return float(np.mean(self.records))

    def sum(self):
        return float(np.sum(self.records))


def compute_time(model: nn.Module, times: int = 64, input_size=(3, 512, 512), batch_size: int = 1):
    """mean eval-forward latency in ms (reference :74-94)."""
    model.eval()
    x = torch.zeros(batch_size, *input_size, device='cuda')
    timer = TicToc()
    with torch.no_grad():
        for it in range(times + 10):
            if it == 10:
                timer.reset()
            timer.tic()
            model(x)
            torch.cuda.synchronize()
            timer.toc()
    return timer.mean() / 1e6
