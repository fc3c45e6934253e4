"""Tensor-level wrappers over the C-ABI (one Python function per entry point of ``include/pqdet_hip.h``).

Used by the public API (``Decode``, ``loss_per_scale``, ``torch_nms``, ``recover_bboxes_prediction``), by the optimiser and
by the per-kernel parity tests.  The training / inference step itself does not go through these: the engine replays a
pre-built ``pq_op`` array with one ``pq_run_ops`` call (:mod:`pqdet_amd.engine`).

Every function requires CUDA(HIP) tensors and raises :class:`pqdet_amd.abi.PQError` on failure -- there is no CPU path.
"""
import ctypes as C
from typing import Optional

import torch

from . import abi
from .abi import check, lib, ptr


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise abi.PQError('pqdet_amd ops need tensors on the GPU (got a %s tensor); there is no CPU fallback'
                              % t.device)


def make_src(x, scale=None, shift=None, ld=None, act=0) -> abi.Src:
    return abi.Src(ptr(x), ptr(scale), ptr(shift), ld if ld is not None else x.shape[-1], act)


def make_gsrc(g, y=None, p=None, q=None, mu=None, r=None, ldg=None, ldy=None) -> abi.GSrc:
    return abi.GSrc(ptr(g), ptr(y), ptr(p), ptr(q), ptr(mu), ptr(r), ldg if ldg is not None else g.shape[-1],
                    (ldy if ldy is not None else (y.shape[-1] if y is not None else 0)))


def make_gsink(g, y=None, scale=None, shift=None, mean=None, invstd=None, partial=None, act=0, accumulate=False,
               finalize=True, ldg=None, ldy=None, ldp=0) -> abi.GSink:
    return abi.GSink(ptr(g), ptr(y), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(partial),
                     ldg if ldg is not None else g.shape[-1], (ldy if ldy is not None else (y.shape[-1] if y is not None else 0)),
                     act, int(accumulate), int(finalize), ldp)


# ----------------------------------------------------------------------------------------------- head decode / loss
def _nchw_view(t) -> abi.HeadView:
    b, c, h, w = t.shape
    assert t.is_contiguous()
    return abi.HeadView(ptr(t), c * h * w, w, 1, h * w)


def _pred_view(t) -> abi.HeadView:
    b, h, w, a, nv = t.shape
    assert t.is_contiguous()
    return abi.HeadView(ptr(t), h * w * a * nv, w * a * nv, a * nv, 1)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, num_classes, stride):
        _need_cuda(raw)
        raw = raw.contiguous().float()
        b, ch, h, w = raw.shape
        nv = 5 + num_classes
        a = ch // nv
        out = torch.empty(b, h, w, a, nv, device=raw.device, dtype=torch.float32)
        args = abi.Decode(_nchw_view(raw), ptr(out), b, h, w, a, num_classes, stride, h * w * a, 0)
        check(lib().pq_decode(C.byref(args), _stream()), 'pq_decode')
        ctx.save_for_backward(out)
        ctx.meta = (b, ch, h, w, a, num_classes, stride)
        return out

    @staticmethod
    def backward(ctx, dpred):
        (pred,) = ctx.saved_tensors
        b, ch, h, w, a, ncls, stride = ctx.meta
        dpred = dpred.contiguous().float()
        draw = torch.empty(b, ch, h, w, device=pred.device, dtype=torch.float32)
        view = abi.HeadView(None, ch * h * w, w, 1, h * w)
        args = abi.DecodeBwd(ptr(pred), ptr(dpred), ptr(draw), view, b, h, w, a, ncls, stride)
        check(lib().pq_decode_bwd(C.byref(args), _stream()), 'pq_decode_bwd')
        return draw, None, None


def decode_nchw(raw: torch.Tensor, num_classes: int, stride: int) -> torch.Tensor:
    """reference ``Decode.forward`` (model/parser.py:206-235): (B, A*(5+C), H, W) -> (B, H, W, A, 5+C)."""
    return _DecodeFn.apply(raw, num_classes, stride)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, boxes, stride, box_kind, ignore_thresh, l1_gain):
        _need_cuda(pred, label, boxes)
        pred = pred.contiguous().float()
        label = label.contiguous().float()
        boxes = boxes.contiguous().float()
        b, h, w, a, nv = pred.shape
        ncls = nv - 5
        m = boxes.shape[1]
        l = lib()
        nblk = l.pq_yolo_loss_blocks(h, w, a)
        partial = torch.empty(b * nblk * 3, device=pred.device, dtype=torch.float32)
        losses = torch.empty(4, device=pred.device, dtype=torch.float32)
        nan_flag = torch.zeros(1, device=pred.device, dtype=torch.int32)
        args = abi.YoloLoss(_pred_view(pred), ptr(label), ptr(boxes), ptr(partial), ptr(losses), ptr(nan_flag), None, None,
                            0, 0, 0, 0, b, h, w, a, ncls, m, stride, box_kind, 0, ignore_thresh, l1_gain)
        check(l.pq_yolo_loss_fwd(C.byref(args), _stream()), 'pq_yolo_loss_fwd')
        ctx.save_for_backward(pred, label, boxes)
        ctx.meta = (stride, box_kind, ignore_thresh, l1_gain)
        ctx.mark_non_differentiable(nan_flag)
        return losses[0:1], losses[1:2], losses[2:3], losses[3:4], nan_flag

    @staticmethod
    def backward(ctx, g_total, g_box, g_conf, g_cls, _g_flag):
        pred, label, boxes = ctx.saved_tensors
        stride, box_kind, ignore_thresh, l1_gain = ctx.meta
        b, h, w, a, nv = pred.shape

        def term(g):
            return g.reshape(1) if g is not None else torch.zeros(1, device=pred.device)
        gt = term(g_total)
        gscale = torch.cat([gt + term(g_box), gt + term(g_conf), gt + term(g_cls)]).float().contiguous()
        grad = torch.empty_like(pred)
        v = _pred_view(pred)
        args = abi.YoloLoss(v, ptr(label), ptr(boxes), None, None, None, ptr(gscale), ptr(grad),
                            v.sb, v.sh, v.sw, v.sc, b, h, w, a, nv - 5, boxes.shape[1], stride, box_kind, 0,
                            ignore_thresh, l1_gain)
        check(lib().pq_yolo_loss_bwd(C.byref(args), _stream()), 'pq_yolo_loss_bwd')
        return grad, None, None, None, None, None, None


def yolo_loss(pred, label, boxes, stride, box_kind, ignore_thresh, l1_gain):
    """-> (loss, box, conf, cls) each shape (1,), plus the device NaN flag (int32[1])."""
    return _LossFn.apply(pred, label, boxes, int(stride), int(box_kind), float(ignore_thresh), float(l1_gain))


def iou_pairs(b1: torch.Tensor, b2: torch.Tensor, kind: int) -> torch.Tensor:
    """Broadcast IoU-family value (reference tools.py:357-477) for boxes (...,4) x (...,4)."""
    _need_cuda(b1, b2)
    shape = torch.broadcast_shapes(b1.shape[:-1], b2.shape[:-1])
    x = b1.float().expand(*shape, 4).contiguous().view(-1, 4)
    y = b2.float().expand(*shape, 4).contiguous().view(-1, 4)
    out = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
    if x.shape[0]:
        check(lib().pq_iou_pairs(ptr(x), ptr(y), ptr(out), x.shape[0], kind, _stream()), 'pq_iou_pairs')
    return out.view(shape)


# ----------------------------------------------------------------------------------------------- post-process
def recover(pred: torch.Tensor, delta: torch.Tensor, ratio: torch.Tensor, orig: torch.Tensor, mutate=True):
    """pred (B,N,5+C); delta (B,2)|(2,) = (dy,dx); ratio (B,)|scalar tensor; orig (B,2)|(2,) = (h,w) -> (B,N,4+C)."""
    _need_cuda(pred)
    assert pred.is_contiguous() and pred.dtype == torch.float32
    b, n, nv = pred.shape
    per_image = int(orig.dim() == 2)
    dev = pred.device
    delta = delta.to(dev, torch.float32).reshape(-1, 2).contiguous()
    ratio = ratio.to(dev, torch.float32).reshape(-1).contiguous()
    orig = orig.to(dev, torch.float32).reshape(-1, 2).contiguous()
    if per_image:
        if delta.shape[0] == 1:
            delta = delta.expand(b, 2).contiguous()
        if ratio.shape[0] == 1:
            ratio = ratio.expand(b).contiguous()
    out = torch.empty(b, n, nv - 1, device=dev, dtype=torch.float32)
    args = abi.Recover(ptr(pred), ptr(delta), ptr(ratio), ptr(orig), ptr(out), b, n, nv - 5, per_image, int(mutate))
    check(lib().pq_recover(C.byref(args), _stream()), 'pq_recover')
    return out


_nms_scratch = {}


_nms_hint = {}     # (N, C) -> candidate capacity that was enough last time


def nms_capacity(n: int, ncls: int) -> int:
    """Candidate capacity for the compacting fast path of ``pq_nms`` (a trained detector passes 10^2..10^4 of the N*C pairs)."""
    return _nms_hint.get((n, ncls), 8192)


def nms_grow(n: int, ncls: int, seen: Optional[int] = None):
    """Called after a ``count == -1`` (capacity exceeded) or with an observed candidate count: adapts the next capacity."""
    cur = nms_capacity(n, ncls)
    want = cur * 4 if seen is None else max(8192, 1 << (int(seen) * 2 - 1).bit_length())
    _nms_hint[(n, ncls)] = min(max(want, 8192), n * ncls)


def nms(bboxes: torch.Tensor, score_thr: float, iou_thr: float, cap: Optional[int] = None, cand_cap: Optional[int] = None):
    """-> (dets[cap,6], count int32[1]) on device; rows beyond count are undefined.  ``cand_cap`` = number of (box, class)
    pairs the sorts are sized for (None: adaptive, 0: all N*C); ``count == -1`` means it was too small (see tools.torch_nms)."""
    _need_cuda(bboxes)
    bboxes = bboxes.contiguous().float()
    n, w = bboxes.shape
    ncls = w - 4
    cap = cap or n * ncls
    if cand_cap is None:
        cand_cap = nms_capacity(n, ncls)
    if cand_cap >= n * ncls:
        cand_cap = 0
    l = lib()
    need = l.pq_nms_scratch_bytes(n, ncls)
    key = (bboxes.device, need)
    scratch = _nms_scratch.get(key)
    if scratch is None:
        scratch = torch.empty(need, device=bboxes.device, dtype=torch.uint8)
        _nms_scratch.clear()
        _nms_scratch[key] = scratch
    dets = torch.empty(cap, 6, device=bboxes.device, dtype=torch.float32)
    count = torch.zeros(1, device=bboxes.device, dtype=torch.int32)
    args = abi.Nms(ptr(bboxes), ptr(dets), ptr(count), ptr(scratch), need, n, ncls, cap, score_thr, iou_thr, cand_cap)
    check(l.pq_nms(C.byref(args), _stream()), 'pq_nms')
    return dets, count


def nms_batch(bboxes: torch.Tensor, score_thr: float, iou_thr: float, cap: Optional[int] = None, cand_cap: Optional[int] = None):
    """Whole batch in ONE ``pq_nms`` call: ``bboxes`` (B, N, 4+C) -> (dets[cap,6], img int32[cap], count int32[1]) on device.
    Every image is suppressed on its own exactly like one ``nms`` call per image (classes and the coordinate offset are per
    image).  ``img`` holds the image of every kept row; the rows come grouped by image with descending scores inside an image
    (capacity <= 8192 candidates per image: one workgroup per image sorts in LDS) or in descending score order across the
    batch (larger capacities: library radix sorts) -- a stable sort by ``img`` gives the per-image lists either way.
    ``count == -1``: ``cand_cap`` (for the whole batch; 8192 per image on the first path) was too small."""
    _need_cuda(bboxes)
    bboxes = bboxes.contiguous().float()
    b, n, w = bboxes.shape
    ncls = w - 4
    total = b * n * ncls
    cap = cap or total
    if cand_cap is None:
        cand_cap = b * nms_capacity(n, ncls)
    if cand_cap >= total:
        cand_cap = 0
    l = lib()
    need = l.pq_nms_batch_scratch_bytes(b, n, ncls)
    key = (bboxes.device, need)
    scratch = _nms_scratch.get(key)
    if scratch is None:
        scratch = torch.empty(need, device=bboxes.device, dtype=torch.uint8)
        _nms_scratch.clear()
        _nms_scratch[key] = scratch
    dets = torch.empty(cap, 6, device=bboxes.device, dtype=torch.float32)
    img = torch.zeros(cap, device=bboxes.device, dtype=torch.int32)
    count = torch.zeros(1, device=bboxes.device, dtype=torch.int32)
    args = abi.Nms(ptr(bboxes), ptr(dets), ptr(count), ptr(scratch), need, n, ncls, cap, score_thr, iou_thr, cand_cap, b, ptr(img))
    check(l.pq_nms(C.byref(args), _stream()), 'pq_nms (batch)')
    return dets, img, count


# ----------------------------------------------------------------------------------------------- misc
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _need_cuda(param, grad)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    args = abi.Adam(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
                    weight_decay, bc1, bc2, grad_scale)
    check(lib().pq_adam_step(C.byref(args), _stream()), 'pq_adam_step')


def sign_l1(grad: torch.Tensor, weight: torch.Tensor, ratio: float):
    """grad += ratio * sign(weight), in place (``pq_sign_l1``)."""
    _need_cuda(grad, weight)
    if not (grad.is_contiguous() and weight.is_contiguous()) or grad.numel() != weight.numel():
        raise ValueError('sign_l1 needs contiguous tensors of equal size')
    check(lib().pq_sign_l1(ptr(grad), ptr(weight), grad.numel(), float(ratio), _stream()), 'pq_sign_l1')
    return grad


def create_label(gt: torch.Tensor, counts, out_hw, num_classes: int, strides=(8, 16, 32), anchors=None,
                 iou_threshold: float = 0.3, cap: Optional[int] = None):
    """Batch label assignment on the device (``pq_create_label``).

    gt (B, G, 6) float32 rows x1,y1,x2,y2,class,mix-up weight; counts (B,) int32 number of real rows per image or None.
    -> (label_s, label_m, label_l, boxes_s, boxes_m, boxes_l, box_counts[B,3]); box lists are (B, cap, 4)."""
    _need_cuda(gt)
    gt = gt.contiguous().float()
    b, g, six = gt.shape
    if six != 6:
        raise ValueError('gt rows must be [x1, y1, x2, y2, class, mixw]')
    dev = gt.device
    cap = cap or 3 * g
    if anchors is None:
        anchors = torch.tensor([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198),
                                (373, 326)], dtype=torch.float32)
    anchors = anchors.to(dev, torch.float32).contiguous()
    if counts is not None:
        counts = counts.to(dev, torch.int32).contiguous()
    labels = [torch.empty(b, h, w, 3, 6 + num_classes, device=dev) for h, w in out_hw]
    boxes = [torch.empty(b, cap, 4, device=dev) for _ in range(3)]
    box_counts = torch.zeros(b, 3, device=dev, dtype=torch.int32)
    a = abi.CreateLabel()
    a.gt, a.counts, a.anchors = ptr(gt), ptr(counts), ptr(anchors)
    for s in range(3):
        a.label[s], a.boxes[s] = ptr(labels[s]), ptr(boxes[s])
        a.H[s], a.W[s], a.stride[s] = int(out_hw[s][0]), int(out_hw[s][1]), int(strides[s])
    a.box_counts = ptr(box_counts)
    a.B, a.G, a.C, a.gt_per_grid, a.cap, a.iou_threshold = b, g, num_classes, 3, cap, float(iou_threshold)
    check(lib().pq_create_label(C.byref(a), _stream()), 'pq_create_label')
    return (*labels, *boxes, box_counts)
