"""Step compiler / executor: turns the parsed layer list into a flat array of ``pq_op`` kernel descriptors per
(input shape, mode) and replays it with ONE ``pq_run_ops`` call for the forward and ONE for the backward.

MI355X-first design notes (see DESIGN.md for the long form)
* activations are NHWC fp32, kept RAW (pre-BatchNorm); the BN affine + activation of a producer is applied by every
  consumer while it loads ("virtual tensors"), so train-mode BN costs no extra pass over the activation;
* BN batch statistics are emitted by the producing conv kernel as per-block partial sums and finalised by a tiny kernel
  (deterministic, no float atomics);
* backward keeps one gradient buffer per tensor; each consumer's dgrad kernel accumulates into it and the last one
  applies act' and emits the BN-backward sums in its epilogue; the BN backward itself is the closed form
  dy = p*g + q*(y-mu) + r applied on load by the producer's wgrad / dgrad kernels;
* parameters and gradients live in two flat buffers (``nn.Parameter.data`` are views), so Adam is one launch and the
  data-parallel gradient exchange is a handful of large RCCL all-reduces over contiguous ranges;
* the step never synchronises with the host: loss scalars stay on the device, the NaN trap is polled one step late.
"""
import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import abi
from .abi import check, lib, ptr
from .model.parser import act_code_of

ALIGN = 16           # floats; every per-channel vector / parameter starts on a 64-byte boundary


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _one(v) -> int:
    """kernel_size / stride / padding of a torch pooling module: int or (h, w) with h == w."""
    return int(v[0] if isinstance(v, (tuple, list)) else v)


class FlatParams:
    """All parameters of a model as views into one flat fp32 buffer (+ a parallel flat gradient buffer)."""

    def __init__(self, model: nn.Module, device):
        params = [p for p in model.parameters()]
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += _round_up(p.numel(), ALIGN)
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.grad = torch.zeros(total, device=device, dtype=torch.float32)
        self.params, self.offsets = params, offs
        self.views, self.grad_views = [], []
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
                self.views.append(v)
                self.grad_views.append(self.grad[o:o + p.numel()].view(p.shape))
                p._pq_flat = (self, o)
        self.index = {id(p): i for i, p in enumerate(params)}

    def still_valid(self) -> bool:
        return all(p.data.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def grad_ptr(self, p) -> int:
        return self.grad_views[self.index[id(p)]].data_ptr()

    def assign_grads(self):
        for p, gv in zip(self.params, self.grad_views):
            if p.grad is None:
                p.grad = gv
            elif p.grad.data_ptr() != gv.data_ptr():
                p.grad.add_(gv)


class _Coef:
    """Arena of small per-channel fp32 vectors (16-float aligned, zero padded)."""

    def __init__(self, device):
        self.device = device
        self.req: List[Tuple[object, str, int]] = []
        self.buf = None

    def want(self, owner, name: str, n: int):
        self.req.append((owner, name, n))

    def commit(self):
        total, offs = 0, []
        for _, _, n in self.req:
            offs.append(total)
            total += _round_up(n, ALIGN) + ALIGN
        self.buf = torch.zeros(max(total, ALIGN), device=self.device, dtype=torch.float32)
        for (owner, name, n), o in zip(self.req, offs):
            setattr(owner, name, self.buf[o:o + n])


class BnState:
    def __init__(self, module: nn.BatchNorm2d, C_: int):
        self.module = module
        self.C = C_
        self.count = 0.0
        self.fwd_nblk = 0
        self.bwd_nblk = 0
        self.ldp = _round_up(C_, ALIGN)
        self.fwd_partial = None
        self.bwd_partial = None
        # coefficient vectors are attached by _Coef.commit(): mean, invstd, scale, shift, p, q, r


class TensorRef:
    """A layer output: raw NHWC buffer (+ BN affine + activation applied by consumers)."""

    def __init__(self, raw, B, H, W, C_, ld, bn: Optional[BnState], act: int):
        self.raw, self.B, self.H, self.W, self.C, self.ld, self.bn, self.act = raw, B, H, W, C_, ld, bn, act
        self.grad = None
        self.n_consumers = 0
        self.received = 0
        self.needs_grad = True

    @property
    def rows(self):
        return self.B * self.H * self.W

    def src(self) -> abi.Src:
        fused = getattr(self, 'fuse_inner', None)
        if fused is not None:          # QAT: read the producer's raw tensor, fake-quantise while loading
            s = fused.src()
            s.qstate = self.fuse_qptr
            return s
        if self.bn is not None:
            return abi.Src(ptr(self.raw), ptr(self.bn.scale), ptr(self.bn.shift), self.ld, self.act)
        return abi.Src(ptr(self.raw), None, None, self.ld, self.act)

    def gsrc(self, bn_train: bool) -> abi.GSrc:
        if self.bn is not None:
            b = self.bn
            return abi.GSrc(ptr(self.grad), ptr(self.raw), ptr(b.p), ptr(b.q), ptr(b.mean), ptr(b.r), self.ld, self.ld)
        return abi.GSrc(ptr(self.grad), None, None, None, None, None, self.ld, 0)


class Plan:
    """Compiled forward (+ backward) op arrays for one input shape and mode."""

    def __init__(self, model, x_shape, bn_train: bool, with_loss: bool):
        self.model = model
        self.bn_train = bn_train
        self.with_loss = with_loss
        self.device = model._flat.flat.device
        self.B, _, self.IH, self.IW = x_shape
        self.fwd: List[abi.Op] = []
        self.bwd: List[abi.Op] = []
        self.keep = []                 # tensors that must stay alive
        self.patch_x: List[Tuple[str, int, str]] = []       # (which, op index, field path)
        self.loss_ops_f: List[Tuple[int, int]] = []          # (fwd op index, head slot)
        self.loss_ops_b: List[Tuple[int, int]] = []
        self.segments: List[Tuple[int, int, int]] = []       # backward: (op_end, flat_lo, flat_hi) gradient-ready marks
        # quantisation-aware training (tools.prepare_qat): observers / fake-quantisation as extra ops around the same kernels
        self.qat = bool(getattr(model, '_qat', False))
        self.q_observe = int(bool(getattr(model, 'qat_observer_enabled', True)))
        self.q_quant = int(bool(getattr(model, 'qat_fake_quant_enabled', True)))
        self.post_bwd: List[abi.Op] = []
        self._inner: Dict[int, TensorRef] = {}                # QAT: layer index -> un-quantised producer tensor
        self._wq: Dict[int, tuple] = {}                       # QAT: layer index -> (wq, sf, dwq) buffers
        self._wpad: Dict[int, torch.Tensor] = {}              # dense k x k convs: weight rows padded to the patch pitch
        self._build()

    # ------------------------------------------------------------------------------------------ helpers
    def _new(self, *shape, fill=None):
        t = torch.empty(*shape, device=self.device, dtype=torch.float32) if fill is None else \
            torch.full(shape, fill, device=self.device, dtype=torch.float32)
        self.keep.append(t)
        return t

    def _bn_scratch(self, C_):
        """(scratch, tickets) shared by every BN finalize that runs on the MAIN stream (issued in order, so one set serves them
        all).  Finalizes of the two-stream forward's side layers run concurrently with them and therefore never get this
        scratch: _emit_forward clears their pointers (single-block path, their partial lists are short)."""
        l = lib()
        need, nt = l.pq_bn_finalize_scratch_bytes(C_), l.pq_bn_finalize_tickets(C_)
        cur = getattr(self, '_bnfin', None)
        if cur is None or cur[0].numel() * 8 < need or cur[1].numel() < nt:
            cmax = max(C_, 2048)
            need, nt = l.pq_bn_finalize_scratch_bytes(cmax), l.pq_bn_finalize_tickets(cmax)
            sc = torch.empty(need // 8, device=self.device, dtype=torch.float64)
            tk = torch.zeros(nt, device=self.device, dtype=torch.int32)
            self.keep += [sc, tk]
            self._bnfin = cur = (sc, tk)
        return cur[0].data_ptr(), cur[1].data_ptr()

    def _qstate(self, key) -> int:
        """Device address of the observer state {min, max, scale, zero_point} of ('in',) / ('w', i) / ('a', i)."""
        m = self.model
        return m._qat_state.data_ptr() + 16 * m._qat_index[key]

    def _qact_fwd(self, src: abi.Src, out_t, rows, C_, ldo, key, flat=0) -> int:
        elems = rows if flat else rows * C_
        part = self._new(2 * lib().pq_qact_blocks(elems) + 8)
        a = abi.Qact(src, ptr(out_t), self._qstate(key), ptr(part), rows, C_, ldo, flat, self.q_observe, self.q_quant, 0.01)
        return self._op(self.fwd, abi.OP_QACT_FWD, a)

    def _qact_bwd(self, qout: 'TensorRef', inner: 'TensorRef', key):
        """Gradient of the quantised tensor -> gradient of its producer (straight-through mask x act', BN-backward sums)."""
        l = lib()
        Cq = inner.C if inner.C % 4 == 0 else inner.ld          # heads: 75 channels in a pitch of 76 (pad column is zero)
        sk = self._sink(inner, l.pq_ew_blocks(inner.rows, Cq))
        a = abi.QactBwd(ptr(qout.grad), sk, self._qstate(key), inner.rows, Cq, qout.ld, self.q_quant)
        self._op(self.bwd, abi.OP_QACT_BWD, a)

    def _op(self, lst, code, args) -> int:
        op = abi.Op()
        op.code = code
        setattr(op.u, abi.OP_FIELD[code], args)
        lst.append(op)
        return len(lst) - 1

    def _sink(self, t: TensorRef, nblk: int, addend_ok: bool = True) -> abi.GSink:
        inner = getattr(t, 'fuse_inner', None)
        if inner is not None:
            # QAT, conv output consumed only through gradient sinks: the consumers write straight into the producer's
            # gradient; the last one applies act' x the quantiser's straight-through mask and emits the BN-backward sums
            if inner.grad is None:
                inner.grad = self._new(inner.rows, inner.ld, fill=0.0) if inner.ld != inner.C else self._new(inner.rows, inner.ld)
                t.grad = inner.grad
            accumulate = t.received > 0
            finalize = (t.received + 1 == t.n_consumers)
            t.received += 1
            bn = inner.bn
            stats = finalize and bn is not None
            if stats:
                bn.bwd_nblk = nblk
                bn.bwd_partial = self._new(nblk * 2 * bn.ldp)
            if finalize:
                inner.received = inner.n_consumers
            return abi.GSink(ptr(inner.grad), ptr(inner.raw), ptr(bn.scale) if bn else None, ptr(bn.shift) if bn else None,
                             ptr(bn.mean) if stats else None, ptr(bn.invstd) if stats else None,
                             ptr(bn.bwd_partial) if stats else None, inner.ld, inner.ld, inner.act, int(accumulate),
                             int(finalize), bn.ldp if bn else 0, self._qstate(t.fuse_key) if self.q_quant else None)
        pend = getattr(t, 'addend', None)
        if pend is not None and not addend_ok:
            self._flush_addend(t)
            pend = None
        if t.grad is None:
            t.grad = self._new(t.rows, t.ld, fill=0.0) if t.ld != t.C else self._new(t.rows, t.ld)
        accumulate = t.received > 0
        finalize = (t.received + 1 == t.n_consumers)
        t.received += 1
        # folded linear shortcut (see _fold_shortcuts): t = a' + b' with a = conv + BatchNorm, no activation.  dL/da' IS dL/dt, so a's
        # gradient is t's buffer and the sink that finalises t emits a's BatchNorm-backward sums (y = a's raw output)
        ys = getattr(t, 'fold_sa', None) or t
        stats = finalize and ys.bn is not None          # frozen statistics still need the sums: they ARE dgamma / dbeta
        if stats:
            b = ys.bn
            b.bwd_nblk = nblk
            b.bwd_partial = self._new(nblk * 2 * b.ldp)
        if ys is not t:
            ys.grad = t.grad
            if finalize:
                ys.received = ys.n_consumers
        bn = ys.bn
        sk = abi.GSink(ptr(t.grad), ptr(ys.raw), ptr(bn.scale) if bn else None, ptr(bn.shift) if bn else None,
                       ptr(bn.mean) if stats else None, ptr(bn.invstd) if stats else None,
                       ptr(bn.bwd_partial) if stats else None, t.ld, ys.ld, ys.act, int(accumulate), int(finalize),
                       bn.ldp if bn else 0)
        if pend is not None:             # the first contribution lives in another tensor's gradient: read it there, write here
            sk.addend = ptr(pend)
            t.addend = None
        return sk

    def _flush_addend(self, t: TensorRef):
        """A pending addend (folded shortcut) in front of a sink that cannot read it from another tensor: copy it."""
        if t.grad is None:
            t.grad = self._new(t.rows, t.ld, fill=0.0) if t.ld != t.C else self._new(t.rows, t.ld)
        sk = abi.GSink(ptr(t.grad), ptr(t.raw), None, None, None, None, None, t.ld, t.ld, 0, 0, 0, 0)
        self._op(self.bwd, abi.OP_ROUTE_BWD, abi.RouteBwd(ptr(t.addend), sk, t.B, t.H, t.W, t.C, t.ld, 0, 1))
        t.addend = None

    # ------------------------------------------------------------------------------------------ build
    def _build(self):
        model, l = self.model, lib()
        mods = list(model.module_list)
        flat = model._flat
        B = self.B
        coef = _Coef(self.device)
        # ---- pass 0: shapes + consumer counts -------------------------------------------------------
        specs = []
        h, w, c = self.IH, self.IW, 3
        shapes: List[Tuple[int, int, int]] = []
        consumers: Dict[int, int] = {}

        def use(idx):
            consumers[idx] = consumers.get(idx, 0) + 1

        alias: Dict[int, int] = {}

        def resolve(i, where):
            j = i if i >= 0 else where + i
            while j in alias:
                j = alias[j]
            return j

        for i, m in enumerate(mods):
            t = m._type
            if t == 'convolutional':
                conv = m.conv
                k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
                if i > 0:
                    use(resolve(i - 1, i))
                h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
                c = conv.out_channels
            elif t == 'shortcut':
                use(resolve(i - 1, i))
                use(resolve(m._from, i))
            elif t == 'route':
                srcs = [resolve(j, i) for j in m._layers]
                if m.single or len(srcs) == 1:     # a one-source route is the source itself (cat of one tensor)
                    alias[i] = srcs[0]
                    h, w, c = shapes[srcs[0]]
                else:
                    for sidx in srcs:
                        use(sidx)
                    h, w = shapes[srcs[0]][0], shapes[srcs[0]][1]
                    c = sum(shapes[sidx][2] for sidx in srcs)
            elif t == 'upsample':
                use(resolve(i - 1, i))
                f = int(m.scale_factor)
                h, w = h * f, w * f
            elif t == 'yolo':
                use(resolve(i - 1, i))
            elif t == 'avgpool':                               # nn.AdaptiveAvgPool2d (global: squeeze of an SE block / classifier head)
                use(resolve(i - 1, i))
                osz = m.output_size if isinstance(m.output_size, (tuple, list)) else (m.output_size, m.output_size)
                h, w = int(osz[0]), int(osz[1])
            elif t == 'maxpool':                               # reference model/parser.py:441-444
                use(resolve(i - 1, i))
                k_, s_, p_ = _one(m.kernel_size), _one(m.stride), _one(m.padding)
                h, w = (h + 2 * p_ - k_) // s_ + 1, (w + 2 * p_ - k_) // s_ + 1
            elif t == 'fc':                                    # reference :122-137: nn.Linear on x.view(input)
                use(resolve(i - 1, i))
                if B * h * w * c != m.input:
                    raise ValueError('fc layer %d expects %d inputs (x.view(input)), the tensor in front of it has %d x %d x %d x %d'
                                     % (i, m.input, B, c, h, w))
                h, w, c = 1, 1, m.fc.out_features
            elif t == 'scale_channels':                        # excitation: feature map x sigmoid(gate)
                use(resolve(i - 1, i))
                use(resolve(m._from, i))
                h, w, c = shapes[resolve(m._from, i)]
            else:
                raise NotImplementedError("layer type '%s' is not supported by the HIP engine yet" % t)
            shapes.append((h, w, c))

        # ---- inverted-residual fusion (csrc/irblock.hip, irbwd.hip): a 1x1 expand conv + BN + ReLU6 whose ONLY consumer is the next
        # layer, a depthwise 3x3 + BN (MobileNetV2's expand -> depthwise pair), runs as pq_xstat_fwd + pq_xdw_fwd in the forward and
        # pq_xdw_bwd in the backward: the expanded tensor and its gradient are never materialised.  Train-mode plans with a loss
        # only (the fused backward folds the train-mode BatchNorm backward analytically), never QAT; PQ_IRFUSE=0 switches it off.
        self.ir_fused: Dict[int, int] = {}                     # depthwise layer index -> its expand layer index
        if self.with_loss and self.bn_train and not self.qat and os.environ.get('PQ_IRFUSE', '1') != '0':
            hw_ = [(self.IH, self.IW)]
            for i, m in enumerate(mods[:-1]):
                d = mods[i + 1]
                if i == 0 or m._type != 'convolutional' or d._type != 'convolutional' or not hasattr(m, 'bn') or not hasattr(d, 'bn'):
                    continue
                ce, cd_ = m.conv, d.conv
                if not (ce.kernel_size[0] == 1 and ce.stride[0] == 1 and ce.groups == 1 and ce.bias is None and ce.padding[0] == 0):
                    continue
                E_, I_ = ce.out_channels, ce.in_channels
                if not (cd_.kernel_size[0] == 3 and cd_.groups == E_ and cd_.in_channels == E_ and cd_.out_channels == E_ and
                        cd_.padding[0] == 1 and cd_.bias is None and cd_.stride[0] in (1, 2)):
                    continue
                if act_code_of(getattr(m, 'act', None)) != abi.ACT_RELU6 or consumers.get(i, 0) != 1 or resolve(i, i + 1) != i:
                    continue
                sidx = resolve(i - 1, i)
                sm = mods[sidx]
                # the block input must be a linear operand: a conv + BN without activation, or a materialised shortcut sum
                if sm._type == 'convolutional':
                    if act_code_of(getattr(sm, 'act', None)) != abi.ACT_LINEAR or not hasattr(sm, 'bn'):
                        continue
                elif sm._type != 'shortcut' or act_code_of(sm.act) != abi.ACT_LINEAR:
                    continue
                H_, W_ = shapes[i][0], shapes[i][1]
                if l.pq_xdw_fwd_blocks(B, H_, W_, I_, E_, cd_.stride[0]) <= 0 or l.pq_xdw_bwd_blocks(B, H_, W_, I_, E_, cd_.stride[0]) <= 0:
                    continue
                self.ir_fused[i + 1] = i
        ir_expand = set(self.ir_fused.values())
        self.ir_expand = ir_expand

        # ---- pass 1: forward ops ---------------------------------------------------------------------
        T: List[Optional[TensorRef]] = []
        self.tensors = T
        heads = []
        train_graph = self.with_loss

        def conv_bn(m, C_out, rows) -> Optional[BnState]:
            if not hasattr(m, 'bn'):
                return None
            b = BnState(m.bn, C_out)
            b.count = float(rows)
            for name in ('mean', 'invstd', 'scale', 'shift', 'p', 'q', 'r'):
                coef.want(b, name, C_out)
            return b

        pending_bn_ops = []   # (op list, index) whose coefficient pointers are filled after coef.commit()
        self._deferred = []

        for i, m in enumerate(mods):
            t = m._type
            if t == 'convolutional':
                conv = m.conv
                k, s = conv.kernel_size[0], conv.stride[0]
                Cin, Cout, groups = conv.in_channels, conv.out_channels, conv.groups
                OH, OW, _ = shapes[i]
                rows = B * OH * OW
                ld = _round_up(Cout, 4)
                bn = conv_bn(m, Cout, rows)
                act = act_code_of(getattr(m, 'act', None))
                is_gate = False
                if act == abi.ACT_LOGISTIC and i + 1 < len(mods) and mods[i + 1]._type == 'scale_channels' \
                        and consumers.get(i, 0) == 1 and not hasattr(m, 'bn'):
                    act, is_gate = 0, True          # the sigmoid of an SE gate is applied inside the scale_channels kernels
                special_act = act in (abi.ACT_LOGISTIC, abi.ACT_TANH)
                if special_act and (train_graph or self.qat):
                    raise NotImplementedError('logistic / tanh as a conv activation have no backward / QAT support (forward only; '
                                              'the gate of a scale_channels layer is the exception)')
                if i in ir_expand:
                    # fused expand layer: no tensor (the depthwise kernel recomputes its values), only the BatchNorm vectors, which
                    # pq_xstat_fwd derives from the moments of the block input (kept in fp64 for the backward)
                    src = T[resolve(i - 1, i)]
                    out = TensorRef(None, B, OH, OW, Cout, ld, bn, act)
                    out.is_gate = False
                    out.ir_src = src
                    out.ir_moments = torch.zeros(Cin * Cin + Cin, device=self.device, dtype=torch.float64)
                    out.ir_scratch = self._new(l.pq_xstat_fwd_scratch_bytes(rows, Cin) // 4 + 16, fill=0.0)
                    self.keep.append(out.ir_moments)
                    bn.fwd_nblk = 0
                    self._deferred.append(('xstat', i, out, src))
                    T.append(out)
                    continue
                raw = self._new(rows, ld) if ld == Cout else self._new(rows, ld, fill=0.0)
                out = TensorRef(raw, B, OH, OW, Cout, ld, bn, act)
                out.is_gate = is_gate
                want_stats = bn is not None and self.bn_train
                if self.qat and Cout % 4 and bn is not None:
                    raise NotImplementedError('quantisation-aware training needs channel counts that are multiples of 4 '
                                              '(layer %d has %d); fine-tune the pruned net in float' % (i, Cout))
                if self.qat:
                    wq = self._new(conv.weight.numel())
                    sfv = self._new(_round_up(Cout, ALIGN), fill=1.0)
                    dwq = self._new(conv.weight.numel()) if train_graph else None
                    self._wq[i] = (wq, sfv, dwq)
                pad = conv.padding[0]
                stem_ok = i == 0 and k == 3 and groups == 1 and Cin == 3 and pad == 1
                c1_ok = i > 0 and k == 1 and groups == 1 and pad == 0
                dw_ok = i > 0 and k == 3 and groups == Cin == Cout and pad == 1 and conv.bias is None     # (no bias input: never dropped silently)
                gc_ok = (i > 0 and k == 3 and pad == 1 and Cin == Cout and Cin % groups == 0 and (Cin // groups) % 4 == 0 and
                         Cin // groups <= 32 and s in (1, 2) and conv.bias is None)
                if not (stem_ok or c1_ok or dw_ok or gc_ok):
                    # any other dense convolution (7 x 7 stem, 3 x 3 with Cin != Cout or wide groups: the classifier cfgs): patch matrix +
                    # the 1x1 GEMM kernels.  Forward only (the reference has no training path for these nets).
                    if groups != 1:
                        raise NotImplementedError('conv k=%d groups=%d (%d->%d, bias=%s): grouped / depthwise convs need batch_normalize=1 '
                                                  '(no bias), equal channel counts and a group width of 1 or a multiple of 4 up to 32'
                                                  % (k, groups, Cin, Cout, conv.bias is not None))
                    if train_graph or self.qat:
                        raise NotImplementedError('dense %d x %d conv %d->%d (layer %d) is forward only (im2col + GEMM): no backward / '
                                                  'QAT support' % (k, k, Cin, Cout, i))
                    K = Cin * k * k
                    ldk = _round_up(K, 4)
                    patches = TensorRef(self._new(rows, ldk), B, OH, OW, ldk, ldk, None, 0)
                    srcT = None if i == 0 else T[resolve(i - 1, i)]
                    self._deferred.append(('lay', i, abi.LAYER_IM2COL, srcT, patches, dict(k=k, stride=s, pad=pad, C=Cin)))
                    if ldk != K:                    # weight rows to the patch pitch (refreshed every forward: the weights may change)
                        wpad = self._new(Cout * ldk)
                        self._wpad[i] = wpad
                        self._deferred.append(('lay', i, abi.LAYER_PADROWS, None, wpad, dict(w=conv.weight, O=Cout, k=K, ldo=ldk)))
                    nblk = l.pq_conv1x1_blocks_k(rows, Cout, ldk, 0)
                    self._deferred.append(('c1_fwd', i, out, nblk, patches, 1))
                elif i == 0:
                    nblk = l.pq_ew_blocks(rows, Cout)
                    if self.qat:
                        self.qimg = self._new(B * 3 * self.IH * self.IW)
                        self._deferred.append(('qimg',))
                    self._deferred.append(('stem_fwd', i, out, nblk))
                else:
                    src = T[resolve(i - 1, i)]
                    if k == 1 and groups == 1:
                        nblk = l.pq_conv1x1_blocks_k(rows, Cout, Cin, 0)
                        self._deferred.append(('c1_fwd', i, out, nblk, src, s))
                    elif i in self.ir_fused:
                        nblk = l.pq_xdw_fwd_blocks(B, src.H, src.W, src.ir_src.C, Cout, s)
                        self._deferred.append(('xdw_fwd', i, out, nblk, src, s))
                    elif k == 3 and groups == Cin == Cout and conv.padding[0] == 1:
                        nblk = l.pq_dw3x3_fwd_blocks(B, OH, OW, Cout, s)
                        self._deferred.append(('dw_fwd', i, out, nblk, src, s))
                    elif (k == 3 and conv.padding[0] == 1 and Cin == Cout and Cin % groups == 0 and
                          (Cin // groups) % 4 == 0 and Cin // groups <= 32 and s in (1, 2) and conv.bias is None):
                        nblk = l.pq_gconv3x3_blocks(B, OH, OW)
                        self._deferred.append(('gc_fwd', i, out, nblk, src, s))
                    else:
                        raise RuntimeError('internal: conv dispatch')
                if bn is not None:
                    bn.fwd_nblk = nblk
                    if want_stats:
                        bn.fwd_partial = self._new(nblk * 2 * bn.ldp)
                    self._deferred.append(('bn_fwd', i, out))
                if self.qat:
                    out.n_consumers = 1
                    self._inner[i] = out
                    qbuf = self._new(rows, ld) if ld == Cout else self._new(rows, ld, fill=0.0)
                    qout = TensorRef(qbuf, B, OH, OW, Cout, ld, None, 0)
                    self._deferred.append(('qact', i, out, qout))
                    out = qout
                if special_act and not is_gate:
                    # consumers only apply the piecewise-linear family while loading: act(bn(conv)) is materialised once
                    mat = TensorRef(self._new(rows, ld), B, OH, OW, Cout, ld, None, 0)
                    self._deferred.append(('lay', i, abi.LAYER_ACT, out, mat, {}))
                    out = mat
                T.append(out)
            elif t == 'maxpool' or (t == 'avgpool' and (tuple(shapes[i][:2]) != (1, 1) or T[resolve(i - 1, i)].C % 4)) or t == 'fc':
                if train_graph or self.qat:
                    raise NotImplementedError("layer type '%s' is forward only (no backward / QAT support)" % t)
                a = T[resolve(i - 1, i)]
                OH, OW, Co = shapes[i]
                if t == 'fc':
                    ld = _round_up(Co, 4)
                    out = TensorRef(self._new(1, ld, fill=0.0), 1, 1, 1, Co, ld, None, 0)
                    self._deferred.append(('lay', i, abi.LAYER_FC, a, out, dict(w=m.fc.weight, bias=m.fc.bias, O=Co,
                                                                                act=act_code_of(m.act))))
                else:
                    out = TensorRef(self._new(B * OH * OW, a.ld, fill=0.0), B, OH, OW, a.C, a.ld, None, 0)
                    if t == 'maxpool':
                        self._deferred.append(('lay', i, abi.LAYER_MAXPOOL, a, out, dict(k=_one(m.kernel_size), stride=_one(m.stride),
                                                                                          pad=_one(m.padding))))
                    else:
                        self._deferred.append(('lay', i, abi.LAYER_AVGPOOL, a, out, {}))
                T.append(out)
            elif t == 'shortcut':
                a, b_ = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                raw = self._new(a.rows, a.ld) if a.ld == a.C else self._new(a.rows, a.ld, fill=0.0)
                out = TensorRef(raw, a.B, a.H, a.W, a.C, a.ld, None, act_code_of(m.act))
                self._deferred.append(('copy', i, out, a, b_, 0, 1))
                if self.qat:                                   # reference: act(fq(a + b))
                    out.act, out.n_consumers = 0, 1
                    self._inner[i] = out
                    qout = TensorRef(self._new(a.rows, a.ld), a.B, a.H, a.W, a.C, a.ld, None, act_code_of(m.act))
                    self._deferred.append(('qact', i, out, qout))
                    out = qout
                T.append(out)
            elif t == 'route':
                srcs = [T[resolve(j, i)] for j in m._layers]
                if m.single or len(srcs) == 1:
                    T.append(srcs[0])
                else:
                    Ctot = sum(s_.C for s_ in srcs)
                    ld = _round_up(Ctot, 4)
                    raw = self._new(srcs[0].rows, ld) if ld == Ctot else self._new(srcs[0].rows, ld, fill=0.0)
                    out = TensorRef(raw, srcs[0].B, srcs[0].H, srcs[0].W, Ctot, ld, None, 0)
                    off = 0
                    for s_ in srcs:
                        self._deferred.append(('copy', i, out, s_, None, off, 1))
                        off += s_.C
                    if self.qat:                               # FloatFunctional.cat: the concatenation is re-quantised
                        out.n_consumers = 1
                        self._inner[i] = out
                        qout = TensorRef(self._new(out.rows, ld), out.B, out.H, out.W, Ctot, ld, None, 0)
                        self._deferred.append(('qact', i, out, qout))
                        out = qout
                    T.append(out)
            elif t == 'upsample':
                a = T[resolve(i - 1, i)]
                f = int(m.scale_factor)
                raw = self._new(a.rows * f * f, a.ld) if a.ld == a.C else self._new(a.rows * f * f, a.ld, fill=0.0)
                out = TensorRef(raw, a.B, a.H * f, a.W * f, a.C, a.ld, None, 0)
                self._deferred.append(('copy', i, out, a, None, 0, f))
                T.append(out)
            elif t == 'avgpool':
                if self.qat:
                    raise NotImplementedError('avgpool has no QAT support')
                a = T[resolve(i - 1, i)]
                if a.C % 4:
                    raise NotImplementedError('avgpool needs a channel count that is a multiple of 4')
                out = TensorRef(self._new(a.B, a.ld), a.B, 1, 1, a.C, a.ld, None, 0)
                self._deferred.append(('pool', i, out, a))
                T.append(out)
            elif t == 'scale_channels':
                gate, feat = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                if not getattr(gate, 'is_gate', False) or gate.C != feat.C or gate.H != 1 or gate.W != 1 or feat.C % 4:
                    raise NotImplementedError('scale_channels needs a (B, C, 1, 1) logistic conv as its gate and C % 4 == 0')
                out = TensorRef(self._new(feat.rows, feat.ld), feat.B, feat.H, feat.W, feat.C, feat.ld, None, 0)
                self._deferred.append(('scale', i, out, feat, gate))
                T.append(out)
            elif t == 'yolo':
                a = T[resolve(i - 1, i)]
                if a.bn is not None or a.act != 0:
                    raise NotImplementedError('yolo layer must follow a linear conv without batch norm')
                heads.append((i, m, a))
                T.append(a)
        if train_graph and not heads:
            raise NotImplementedError('a net without yolo layers has no loss: forward only (the reference never trains AnyModel)')
        for i_, n_ in consumers.items():
            T[i_].n_consumers = n_
        self._consumers = dict(consumers)
        # Folded backward of LINEAR shortcuts (mobilenetv2: t = a' + b', a = project conv + BatchNorm, no activation on either):
        # dL/da' = dL/db' = dL/dt.  Instead of a pass that copies dL/dt into a's gradient (with a's BatchNorm-backward sums) and adds
        # it to b's (pq_route_bwd with two sinks: four tensor passes and a launch per block on the main chain), a's gradient IS t's
        # buffer -- the sink that finalises t emits a's sums (_sink) -- and b's finalising sink reads its first contribution from t's
        # gradient (pq_gsink.addend).  PQ_SCFOLD=0 switches it off.
        self.sc_folded = self.sc_aliased = 0
        if train_graph and not self.qat and os.environ.get('PQ_SCFOLD', '1') != '0':
            head_in = {id(a_) for _, _, a_ in heads}             # (a loss layer writes its input's gradient itself, not through a sink)
            for i_, m_ in enumerate(mods):
                if m_._type != 'shortcut':
                    continue
                out_, sa_, sb_ = T[i_], T[resolve(i_ - 1, i_)], T[resolve(m_._from, i_)]
                if id(out_) in head_in or id(sa_) in head_in or id(sb_) in head_in:
                    continue
                if out_.bn is None and sa_ is not sb_ and sa_.bn is not None and sa_.act == 0 and \
                        sa_.n_consumers == 1 and sa_.ld == out_.ld and sa_.C % 4 == 0 and out_.n_consumers > 0 and \
                        mods[resolve(i_ - 1, i_)]._type == 'convolutional':
                    if out_.act == 0:
                        out_.fold_sa = sa_
                        self.sc_folded += 1
                    elif out_.act in (1, 2, 3):
                        # an activation behind the sum (RegNet: ReLU): the sum's sinks need the sum's own y for act', so they cannot
                        # emit the conv branch's BatchNorm sums -- but the gradient w.r.t. the raw sum is still the gradient of both
                        # addends: the conv branch uses the sum's buffer, ONE statistics-only pass (pq_route_bwd, finalize = 3: two
                        # reads, no store) replaces the six tensor passes of the two-sink launch, the other addend reads it as addend
                        out_.alias_sa = sa_
                        self.sc_aliased += 1
        if self.qat:
            # conv outputs that only feed other layers are never materialised in quantised form: consumers fake-quantise while
            # loading (pq_src.qstate) and their gradient sinks apply the straight-through mask (pq_gsink.qstate)
            head_inputs = {id(a) for _, _, a in heads}
            for i_, inner in self._inner.items():
                qout = T[i_]
                if mods[i_]._type == 'convolutional' and id(qout) not in head_inputs and qout.act == 0 \
                        and inner.C % 4 == 0 and os.environ.get('PQ_QAT_FUSE', '1') != '0':
                    qout.fuse_inner, qout.fuse_key = inner, ('a', i_)
                    qout.fuse_qptr = self._qstate(('a', i_)) if self.q_quant else None
                    # ... and the observer needs no pass over the tensor either: the producer's epilogue adds per-channel min / max
                    # to its statistic rows (4 planes) and pq_bn_fwd_finalize feeds the moving-average observer from them
                    kinds = {it[0] for it in self._deferred if len(it) > 1 and it[1] == i_}
                    if self.q_observe and self.bn_train and inner.bn is not None and kinds & {'c1_fwd', 'dw_fwd', 'stem_fwd'} \
                            and os.environ.get('PQ_QAT_FOLD', '1') != '0':
                        inner.fold_obs = True
                        inner.bn.fwd_partial = self._new(inner.bn.fwd_nblk * 4 * inner.bn.ldp)
                        inner.fold_mm = torch.tensor([-1, 0, 0, 0], dtype=torch.int32, device=self.device)   # {0xffffffff, 0, 0}
                        self.keep.append(inner.fold_mm)

        coef.commit()
        self.keep.append(coef.buf)
        self._emit_forward(flat, heads)
        if train_graph:
            self._emit_backward(flat, mods, heads, resolve)
        self.fwd_arr = (abi.Op * len(self.fwd))(*self.fwd)
        self.bwd_arr = (abi.Op * len(self.bwd))(*self.bwd) if self.bwd else None
        self.loss_bwd_extra = None
        if self.bwd_arr is not None and getattr(self, '_loss_extra', None):
            # argument structs of the heads whose gradients ride in the first head's launch: host memory the call reads
            self.loss_bwd_extra = (abi.YoloLoss * len(self._loss_extra))(*self._loss_extra)
            first = self.bwd_arr[self._loss_fused_at].u.loss
            for j in range(len(self._loss_extra)):
                first.also[j] = C.addressof(self.loss_bwd_extra[j])
        if self.bwd_arr is not None and self.loss_ops_b:
            self.patch_upstream(self.default_upstream)
        self.post_arr = (abi.Op * len(self.post_bwd))(*self.post_bwd) if self.post_bwd else None

    # ------------------------------------------------------------------------------------------ forward emission
    def _emit_forward(self, flat, heads):
        model, l = self.model, lib()
        mods = list(model.module_list)
        B = self.B
        def wptr(i):
            if i in self._wpad:
                return ptr(self._wpad[i])
            return ptr(self._wq[i][0]) if self.qat else ptr(mods[i].conv.weight)

        if self.qat:
            self._emit_qat_weight_ops(flat, mods)
        # statistics not trained (inference, model.eval() losses, tools.freeze_bn_stats): the per-channel vectors of all
        # BatchNorms come from the running statistics in ONE launch at the start (slot filled in after the loop)
        bne_idx = self._op(self.fwd, abi.OP_BN_EVAL, abi.BnEval(None, 0)) if not self.bn_train else None
        # train-mode BatchNorm statistics handed from producer to first consumer without a finalize launch (_handover_bn_stats):
        # slot 0 clears the accumulators of all such layers (a no-op until the hand-over pass fills it in)
        self._stat_fill_idx = None
        if self.bn_train and not self.qat and os.environ.get('PQ_BN_ATOMIC', '1') != '0':
            self._stat_dummy = self._new(4)
            self._stat_fill_idx = self._op(self.fwd, abi.OP_FILL, abi.Fill(self._stat_dummy.data_ptr(), 0.0, 0))
        bne_items = []
        self.fwd_layer = [-1] * len(self.fwd)      # layer index of every forward op (-1: whole-net ops); tests replay single layers
        cur_layer = -1
        for item in self._deferred:
            kind = item[0]
            self.fwd_layer += [cur_layer] * (len(self.fwd) - len(self.fwd_layer))       # ops of the previous item
            cur_layer = item[1] if len(item) > 1 and isinstance(item[1], int) else -1
            if kind == 'qimg':                                 # QuantStub: observe + fake-quantise the input batch
                idx = self._qact_fwd(abi.Src(None, None, None, 0, 0), self.qimg, B * 3 * self.IH * self.IW, 1, 1, ('in',), flat=1)
                self.patch_x.append(('fwd', idx, 'qa'))
            elif kind == 'qact':
                _, i, inner, qout = item
                if getattr(qout, 'fuse_inner', None) is not None:
                    if self.q_observe and not getattr(inner, 'fold_obs', False):    # observer only; nobody reads a materialised copy
                        self._qact_fwd(inner.src(), None, inner.rows, inner.C, qout.ld, ('a', i))
                else:
                    self._qact_fwd(inner.src(), qout.raw, inner.rows, inner.C, qout.ld, ('a', i))
            elif kind == 'stem_fwd':
                _, i, out, nblk = item
                conv = mods[i].conv
                part = out.bn.fwd_partial if (out.bn and self.bn_train) else None
                a = abi.StemFwd(ptr(self.qimg) if self.qat else None, wptr(i), ptr(out.raw), ptr(part), B, 3, self.IH, self.IW,
                                out.H, out.W, out.C, conv.stride[0], out.ld, out.bn.ldp if out.bn else 0,
                                int(getattr(out, 'fold_obs', False)))
                idx = self._op(self.fwd, abi.OP_STEM_FWD, a)
                if not self.qat:
                    self.patch_x.append(('fwd', idx, 'stf'))
            elif kind == 'c1_fwd':
                _, i, out, nblk, src, s = item
                conv = mods[i].conv
                part = out.bn.fwd_partial if (out.bn and self.bn_train) else None
                a = abi.Conv1x1Fwd(src.src(), wptr(i), ptr(conv.bias) if conv.bias is not None else None,
                                   ptr(out.raw), ptr(part), out.rows, src.C, out.C, out.ld, out.bn.ldp if out.bn else 0,
                                   s, out.H, out.W, src.H, src.W, int(getattr(out, 'fold_obs', False)))
                self._op(self.fwd, abi.OP_CONV1X1_FWD, a)
            elif kind == 'xstat':
                _, i, out, src = item
                b = out.bn
                bm = b.module
                a = abi.XstatFwd(src.src(), wptr(i), ptr(bm.weight), ptr(bm.bias), ptr(bm.running_mean), ptr(bm.running_var),
                                 ptr(bm.num_batches_tracked), ptr(b.mean), ptr(b.invstd), ptr(b.scale), ptr(b.shift),
                                 ptr(out.ir_scratch), out.rows, src.C, out.C, float(bm.eps),
                                 float(bm.momentum if bm.momentum is not None else 0.1), out.ir_moments.data_ptr())
                self._op(self.fwd, abi.OP_XSTAT_FWD, a)
            elif kind == 'xdw_fwd':
                _, i, out, nblk, mid, s = item
                x_in = mid.ir_src
                part = out.bn.fwd_partial if (out.bn and self.bn_train) else None
                a = abi.XdwFwd(x_in.src(), wptr(self.ir_fused[i]), abi.Src(None, ptr(mid.bn.scale), ptr(mid.bn.shift), 0, mid.act), None,
                               wptr(i), ptr(out.raw), ptr(part), B, mid.H, mid.W, out.H, out.W, x_in.C, out.C, s, 0, out.ld,
                               out.bn.ldp if out.bn else 0)
                self._op(self.fwd, abi.OP_XDW_FWD, a)
            elif kind == 'dw_fwd':
                _, i, out, nblk, src, s = item
                conv = mods[i].conv
                part = out.bn.fwd_partial if (out.bn and self.bn_train) else None
                a = abi.DwFwd(src.src(), wptr(i), ptr(out.raw), ptr(part), B, src.H, src.W, out.H, out.W, out.C, s,
                              out.ld, out.bn.ldp if out.bn else 0, int(getattr(out, 'fold_obs', False)))
                self._op(self.fwd, abi.OP_DW_FWD, a)
            elif kind == 'gc_fwd':
                _, i, out, nblk, src, s = item
                conv = mods[i].conv
                part = out.bn.fwd_partial if (out.bn and self.bn_train) else None
                a = abi.GconvFwd(src.src(), wptr(i), ptr(out.raw), ptr(part), B, src.H, src.W, out.H, out.W, out.C,
                                 out.C // conv.groups, s, out.ld, out.bn.ldp if out.bn else 0)
                self._op(self.fwd, abi.OP_GCONV_FWD, a)
            elif kind == 'bn_fwd':
                _, i, out = item
                b = out.bn
                bm = b.module
                train = self.bn_train
                if not train:
                    bne_items.append(abi.BnEvalItem(ptr(bm.weight), ptr(bm.bias), ptr(bm.running_mean), ptr(bm.running_var),
                                                    ptr(b.mean), ptr(b.invstd), ptr(b.scale), ptr(b.shift),
                                                    ptr(self._wq[i][1]) if self.qat else None, float(bm.eps), b.C))
                    continue
                sc, tk = self._bn_scratch(b.C)
                a = abi.BnFwdFin(ptr(b.fwd_partial) if train else None, ptr(bm.weight), ptr(bm.bias), ptr(bm.running_mean),
                                 ptr(bm.running_var), ptr(bm.num_batches_tracked) if train else None, ptr(b.mean),
                                 ptr(b.invstd), ptr(b.scale), ptr(b.shift), b.fwd_nblk, b.ldp, b.C, int(train), b.count,
                                 float(bm.eps), float(bm.momentum if bm.momentum is not None else 0.1), sc, tk,
                                 ptr(self._wq[i][1]) if self.qat else None)
                if getattr(out, 'fold_obs', False):
                    a.planes, a.act, a.qstate, a.mm, a.avg_const = 4, out.act, self._qstate(('a', i)), out.fold_mm.data_ptr(), 0.01
                self._op(self.fwd, abi.OP_BN_FWD_FIN, a)
            elif kind == 'lay':
                _, i, lk, a_, out, kw = item
                la = abi.LayerFwd()
                la.kind = lk
                la.out = ptr(out) if isinstance(out, torch.Tensor) else ptr(out.raw)
                if lk == abi.LAYER_PADROWS:
                    la.w, la.O, la.k, la.ldo = ptr(kw['w']), kw['O'], kw['k'], kw['ldo']
                else:
                    if a_ is None:                             # the network input itself (NCHW): pointer patched per call
                        la.nchw, la.B, la.IH, la.IW, la.C = 1, B, self.IH, self.IW, 3
                    else:
                        la.inp, la.B, la.IH, la.IW, la.C = a_.src(), a_.B, a_.H, a_.W, a_.C
                    la.OH, la.OW, la.ldo = out.H, out.W, out.ld
                    la.k, la.stride, la.pad = kw.get('k', 0), kw.get('stride', 1), kw.get('pad', 0)
                    if lk == abi.LAYER_IM2COL:
                        la.C = kw['C']
                    if lk == abi.LAYER_FC:
                        la.w, la.bias, la.O, la.act = ptr(kw['w']), ptr(kw['bias']), kw['O'], kw['act']
                idx = self._op(self.fwd, abi.OP_LAYER_FWD, la)
                if lk == abi.LAYER_IM2COL and a_ is None:
                    self.patch_x.append(('fwd', idx, 'lay'))
            elif kind == 'pool':
                _, i, out, a_ = item
                pr = abi.PoolReduce(a_.src(), None, 0, None, 0, ptr(out.raw), out.ld, a_.B, a_.H * a_.W, a_.C, 1.0 / (a_.H * a_.W))
                self._op(self.fwd, abi.OP_POOL_REDUCE, pr)
            elif kind == 'scale':
                _, i, out, feat, gate = item
                none = abi.Src(None, None, None, 0, 0)
                a = abi.Copy(feat.src(), none, ptr(out.raw), out.B, out.H, out.W, feat.C, out.ld, 0, 1, ptr(gate.raw), gate.ld)
                self._op(self.fwd, abi.OP_COPY_FWD, a)
            elif kind == 'copy':
                _, i, out, a_, b_, coff, up = item
                none = abi.Src(None, None, None, 0, 0)
                a = abi.Copy(a_.src(), b_.src() if b_ is not None else none, ptr(out.raw), out.B, out.H, out.W, a_.C,
                             out.ld, coff, up)
                self._op(self.fwd, abi.OP_COPY_FWD, a)
        self.fwd_layer += [cur_layer] * (len(self.fwd) - len(self.fwd_layer))
        if bne_idx is not None:
            if bne_items:
                table = (abi.BnEvalItem * len(bne_items))(*bne_items)
                dev_table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
                self.keep.append(dev_table)
                self.fwd[bne_idx].u.bne = abi.BnEval(dev_table.data_ptr(), len(bne_items))
            else:                                   # a net without BatchNorm: turn the slot into a no-op
                dummy = self._new(4)
                nop = abi.Op()
                nop.code = abi.OP_FILL
                nop.u.fill = abi.Fill(dummy.data_ptr(), 0.0, 0)
                self.fwd[bne_idx] = nop
        # heads
        nh = len(heads)
        self.nheads = nh
        if self.with_loss:
            self.loss_buf = self._new(nh, 4, fill=0.0)
            self.loss_total = self._new(4 + nh, fill=0.0)      # loss, box, conf, cls summed over the heads, then each head's total
            # what the backward loss ops read outside a real backward (profiling replays of bwd_arr): d(total loss) = 1
            self._one = self._new(1, fill=1.0)
            self.default_upstream = [self._one] + [None] * (3 + nh)
            # NaN trap (reference model/loss.py:110-114): one sticky flag per head.  In pinned host memory when the runtime maps
            # it into the device's address space (the loss finalize kernels then set the flag the host polls: no device-to-host
            # copy and no event on the stream between forward and backward); a device tensor the model copies back otherwise.
            self.nan_host, nan_ptr = None, None
            if os.environ.get('PQ_NAN_MAPPED', '1') != '0':
                host = torch.zeros(nh, dtype=torch.int32).pin_memory()
                dp = C.c_void_p()
                if l.pq_host_device_pointer(C.c_void_p(host.data_ptr()), C.byref(dp)) == 0 and dp.value:
                    self.nan_host, nan_ptr = host, dp.value
            self.nan_dev = torch.zeros(nh, device=self.device, dtype=torch.int32)
            self.keep.append(self.nan_dev)
            if nan_ptr is None:
                nan_ptr = self.nan_dev.data_ptr()
            # reserved for the weight transposes of the backward (see _emit_backward: a side-stream op in front of the loss ops)
            dummy = self._new(4)
            self._fwd_tr_slot = self._op(self.fwd, abi.OP_FILL, abi.Fill(dummy.data_ptr(), 0.0, 0))
            # ... and for clearing the accumulators of the backward statistics hand-over (_handover_bn_bwd)
            self._fwd_bfill_slot = self._op(self.fwd, abi.OP_FILL, abi.Fill(dummy.data_ptr(), 0.0, 0))
            for slot, (i, m, a) in enumerate(heads):
                opt = m.opt
                A = a.C // (5 + opt['classes'])
                nblk = l.pq_yolo_loss_blocks(a.H, a.W, A)
                partial = self._new(B * nblk * 3)
                view = abi.HeadView(ptr(a.raw), a.H * a.W * a.ld, a.W * a.ld, a.ld, 1)
                if a.grad is None:
                    a.grad = self._new(a.rows, a.ld, fill=0.0)
                la = abi.YoloLoss(view, None, None, ptr(partial), self.loss_buf[slot].data_ptr(),
                                  nan_ptr + 4 * slot, None, ptr(a.grad),
                                  a.H * a.W * a.ld, a.W * a.ld, a.ld, 1,
                                  B, a.H, a.W, A, opt['classes'], 1, opt['stride'], abi.BOX_KIND[opt['bbox_loss']], 1,
                                  float(opt['ignore_thresh']), float(opt['l1_loss_gain']))
                la.head_slot, la.n_heads = slot, nh
                if slot == nh - 1:
                    # the last head: its per-cell kernel only needs its own head conv; its finalize also sums the table (what
                    # DetectionModel.forward returns) and therefore waits for the other heads' loss ops: two ops, so that a
                    # stream join can sit between them (two-stream forward below)
                    la.phase = 1
                    idx = self._op(self.fwd, abi.OP_LOSS_FWD, la)
                    la.phase, la.totals = 2, self.loss_total.data_ptr()
                    self._last_loss_fin = self._op(self.fwd, abi.OP_LOSS_FWD, la)
                    la.phase, la.totals = 0, None
                    self.loss_ops_f.append((self._last_loss_fin, slot, opt['stride']))
                else:
                    idx = self._op(self.fwd, abi.OP_LOSS_FWD, la)
                self.loss_ops_f.append((idx, slot, opt['stride']))
                self._head_loss_args = getattr(self, '_head_loss_args', []) + [(la, slot, opt['stride'], a)]
            # Two-stream forward: the detection-head branches of all but the last level (the layers that only feed their yolo layer:
            # 4096- / 16384-row GEMMs that leave most CUs idle) and the loss kernels run on the side stream beside the next level's
            # neck; pq_run_ops2 forks once per branch and joins at the end of the forward.  Every side layer's output is read by the
            # next side layer only, its input was produced on the main stream before the fork.  11.32 -> 11.24 ms per step
            # (PQ_FWD_SIDE=0: single-stream forward).
            self.fwd_two_streams = False
            if os.environ.get('PQ_FWD_SIDE', '1') != '0' and self.bn_train and len(heads) > 1:
                side_layers = set()
                for i, m, a in sorted(heads, key=lambda h: h[0])[:-1]:
                    j = i - 1
                    # a branch layer's ONLY consumer is the next branch layer (the head conv's: its yolo layer), so nothing on
                    # the main stream ever reads a tensor written on the side stream before the join at the end of the forward
                    while j >= 0 and mods[j]._type == 'convolutional' and self._consumers.get(j, 0) == 1:
                        side_layers.add(j)
                        j -= 1
                self.fwd_layer += [-1] * (len(self.fwd) - len(self.fwd_layer))
                last_kernel, last_fin = self._last_loss_fin - 1, self._last_loss_fin
                for k, layer in enumerate(self.fwd_layer):
                    if k == last_kernel:
                        pass         # the last head's per-cell kernel: main stream, right behind its head conv (no fork latency)
                    elif k == last_fin:
                        # ... its finalize reads the rows the other heads' loss ops wrote on the side stream: behind a join, which
                        # by then has had the whole per-cell kernel to complete
                        self.fwd[k].flags = abi.OPF_JOIN
                    elif layer in side_layers or self.fwd[k].code == abi.OP_LOSS_FWD:
                        self.fwd[k].flags = abi.OPF_SIDE
                        if self.fwd[k].code == abi.OP_BN_FWD_FIN:       # never the main stream's reduction scratch / tickets
                            self.fwd[k].u.bnf.scratch = None
                            self.fwd[k].u.bnf.tickets = None
                self.fwd_two_streams = True
                self.fwd_side_layers = sorted(side_layers)
                self._check_side_forward()
            self._handover_bn_stats()
            return
        elif not heads:
            # a net without yolo layers (classifier cfgs): the output is the last layer's tensor (reference model/interpreter.py:60-65),
            # materialised if it is still virtual
            last = self.tensors[-1]
            if last.bn is not None or last.act != 0:
                mat = TensorRef(self._new(last.rows, last.ld), last.B, last.H, last.W, last.C, last.ld, None, 0)
                la = abi.LayerFwd()
                la.kind, la.out, la.inp = abi.LAYER_ACT, ptr(mat.raw), last.src()
                la.B, la.IH, la.IW, la.C, la.ldo = last.B, last.H, last.W, last.C, mat.ld
                self._op(self.fwd, abi.OP_LAYER_FWD, la)
                last = mat
            self.eval_out = None
            self.final = last
        else:
            rows_total = sum(a.H * a.W * (a.C // (5 + m.opt['classes'])) for _, m, a in heads)
            nv = 5 + heads[0][1].opt['classes']
            self.eval_out = self._new(B, rows_total, nv)
            row0 = 0
            for i, m, a in heads:
                opt = m.opt
                A = a.C // (5 + opt['classes'])
                view = abi.HeadView(ptr(a.raw), a.H * a.W * a.ld, a.W * a.ld, a.ld, 1)
                d = abi.Decode(view, ptr(self.eval_out), B, a.H, a.W, A, opt['classes'], opt['stride'], rows_total, row0)
                self._op(self.fwd, abi.OP_DECODE, d)
                row0 += a.H * a.W * A
        self._handover_bn_stats()

    def _handover_bn_stats(self):
        """Drop the BatchNorm finalize launch between a conv and the FIRST consumer of its output wherever both kernels support
        the hand-over (pq_bn_stat, include/pqdet_hip.h): the producer adds its statistic partials into exact fixed-point
        accumulators, the consumer derives scale / shift in its prologue and its first workgroup stores the vectors (every later
        consumer and the backward read those) and updates the running statistics.  Only on the main stream (a later side-stream
        consumer forks after the first consumer, so it sees the vectors), only short partial lists (same-address atomics
        serialise: scripts/probes/bnstat_probe.hip), never for QAT plans.  The finalize op becomes a no-op (no launch); a
        dependent launch + its dispatch gap cost ~10 us per layer in the step."""
        self.bn_handover = 0
        if self._stat_fill_idx is None:
            return
        l = lib()
        maxblk = 640         # row blocks adding to one channel: beyond that the same-address atomics cost more than the launch
        picks, total = [], 0
        for k in range(1, len(self.fwd) - 1):
            op = self.fwd[k]
            if op.code != abi.OP_BN_FWD_FIN or op.flags:
                continue
            f = op.u.bnf
            if not f.train or f.planes == 4 or f.sf or f.nblk > maxblk or not f.partial:
                continue
            prod, cons = self.fwd[k - 1], self.fwd[k + 1]
            if prod.flags or cons.flags or prod.code not in (abi.OP_CONV1X1_FWD, abi.OP_DW_FWD, abi.OP_GCONV_FWD):
                continue
            pa = getattr(prod.u, abi.OP_FIELD[prod.code])
            if pa.partial != f.partial or not l.pq_bn_stat_ok(prod.code, C.byref(pa), 0):
                continue
            if cons.code == abi.OP_CONV1X1_FWD:
                srcs = [cons.u.c1f.inp]
            elif cons.code == abi.OP_DW_FWD:
                srcs = [cons.u.dwf.inp]
            elif cons.code == abi.OP_GCONV_FWD:
                srcs = [cons.u.gcf.inp]
            elif cons.code == abi.OP_COPY_FWD:
                srcs = [cons.u.cp.a, cons.u.cp.b]
            elif cons.code == abi.OP_XSTAT_FWD:                # the moments pass of a fused block is the first reader of the block input
                srcs = [cons.u.xsf.inp]
            else:
                continue
            hit = [s_ for s_ in srcs if s_.x == pa.out and s_.scale == f.scale and not s_.qstate]
            if len(hit) != 1 or not l.pq_bn_stat_ok(cons.code, C.byref(getattr(cons.u, abi.OP_FIELD[cons.code])), 1):
                continue
            slots = l.pq_bn_stat_slots(f.nblk)
            nbytes = _round_up(l.pq_bn_stat_bytes(f.ldp, slots), 64)
            picks.append((k, pa, hit[0], f, slots, total, total + nbytes))
            total += nbytes + 64                         # + the layer's `bad` word on its own 64-byte line
        if not picks:
            return
        arena = torch.zeros(total // 8, device=self.device, dtype=torch.int64)
        self.keep.append(arena)
        base = arena.data_ptr()
        for k, pa, src, f, slots, off_acc, off_bad in picks:
            pa.partial, pa.stat_acc, pa.stat_bad, pa.stat_slots = None, base + off_acc, base + off_bad, slots
            src.stat = abi.BnStat(base + off_acc, base + off_bad, f.gamma, f.beta, f.running_mean, f.running_var,
                                  f.num_batches_tracked, f.mean, f.invstd, f.scale, f.shift, slots, f.ldp, f.count, f.eps,
                                  f.momentum, 1.0 / f.count)
            nop = abi.Op()
            nop.code = abi.OP_FILL
            nop.u.fill = abi.Fill(self._stat_dummy.data_ptr(), 0.0, 0)
            self.fwd[k] = nop
        self.fwd[self._stat_fill_idx].u.fill = abi.Fill(base, 0.0, total // 4)
        self.bn_handover = len(picks)

    def _check_side_forward(self):
        """Plan-level proof of the two-stream forward's race freedom: every buffer a side op writes is read only by LATER side
        ops (or by the backward, after the join), and no side op shares reduction scratch with a main-stream op."""
        side_written = {}
        for k, op in enumerate(self.fwd):
            side = bool(op.flags & abi.OPF_SIDE)
            a = getattr(op.u, abi.OP_FIELD[op.code])
            reads = []
            if op.code in (abi.OP_CONV1X1_FWD, abi.OP_DW_FWD, abi.OP_GCONV_FWD, abi.OP_QACT_FWD):
                reads.append(a.inp.x)
            elif op.code == abi.OP_COPY_FWD:
                reads += [a.a.x, a.b.x]
            elif op.code == abi.OP_POOL_REDUCE:
                reads.append(a.a.x)
            elif op.code == abi.OP_LOSS_FWD:
                reads.append(a.inp.base)
            for r in reads:
                if r and r in side_written and not side:
                    raise RuntimeError('internal: forward op %d on the main stream reads a tensor written by side-stream op %d'
                                       % (k, side_written[r]))
            if side and op.code == abi.OP_LOSS_FWD and reads and all(r in side_written for r in reads):
                # the loss kernels of a side-stream head read their head conv (side stream), the labels and their own rows: no
                # fork event -- they start when the head conv is done, not when the main stream reaches the end of the forward
                op.flags |= abi.OPF_NOFORK
            if op.flags & abi.OPF_NOFORK and op.code not in (abi.OP_TRANSPOSE, abi.OP_FILL) \
                    and not (side and reads and all(r in side_written for r in reads)):
                raise RuntimeError('internal: forward op %d is flagged NOFORK but reads a tensor the side stream did not write' % k)
            if side:
                out = getattr(a, 'out', None)
                if out:
                    side_written[out] = k
                if op.code == abi.OP_BN_FWD_FIN and (a.scratch or a.tickets):
                    raise RuntimeError('internal: side-stream BatchNorm finalize %d shares the main stream\'s scratch' % k)

    def _emit_qat_weight_ops(self, flat, mods):
        """One table for every conv layer: forward = scale by gamma/sqrt(running_var+eps), observe, fake-quantise (first op
        of the step); backward = gradient of the float weights (+ the BatchNorm-scale term) from the gradient of the
        quantised ones (runs after both streams have joined)."""
        items = []
        for i in sorted(self._wq):
            m = mods[i]
            conv = m.conv
            wq, sfv, dwq = self._wq[i]
            bn = getattr(m, 'bn', None)
            O_ = conv.out_channels
            items.append(abi.QatWeightItem(
                ptr(conv.weight), ptr(bn.weight) if bn is not None else None, ptr(bn.running_var) if bn is not None else None,
                ptr(wq), ptr(sfv), self._qstate(('w', i)), ptr(dwq), flat.grad_ptr(conv.weight),
                flat.grad_ptr(bn.weight) if bn is not None else None, float(bn.eps) if bn is not None else 0.0,
                O_, conv.weight.numel() // O_))
        table = (abi.QatWeightItem * len(items))(*items)
        dev_table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
        self.keep.append(dev_table)
        args = abi.QatWeight(dev_table.data_ptr(), len(items), self.q_observe, self.q_quant, 0.01)
        self._op(self.fwd, abi.OP_QAT_WPREP, args)
        if self.with_loss:
            self._op(self.post_bwd, abi.OP_QAT_WGRAD, abi.QatWeight(dev_table.data_ptr(), len(items), 0, self.q_quant, 0.01))

    # ------------------------------------------------------------------------------------------ backward emission
    def _emit_backward(self, flat, mods, heads, resolve):
        l = lib()
        B = self.B
        T = self.tensors
        # heads first (reverse layer order overall, but head losses only touch their own tensors)
        head_at = {i: (slot, m, a) for slot, (i, m, a) in enumerate(heads)}
        scratch_need = 0
        wg_ops = []
        # Deferred partial-sum reductions (pq_reduce_batch): the weight-gradient entries leave their per-split slabs in their scratch
        # (dw == NULL) and ONE launch per group of layers sums them -- instead of one or two small launches per layer on the
        # weight-gradient stream.  Every conv layer leaves a place-holder slot behind its ops; the slots that become real launches are
        # chosen below (`red_groups` of them), and only the gradient-ready marks behind such a launch survive.
        # MEASURED (round 5, same box, PQ_REDUCE_BATCH=1 against 0): 83 -> 7 reduce launches and 0.39 ms less device time per
        # mobilenetv2-fpn step, and the step is NOT faster: 8.86-8.88 against 8.84 ms; regnetx-600m-fpn 11.33 against 11.16, 960 x 960
        # 16.85 / 16.85, QAT 10.69 / 10.63.  The per-layer reductions read slabs that are still in L2 and fill gaps of the
        # weight-gradient stream; a launch that sums the slabs of ten layers reads them from HBM beside the main chain.  Off by default.
        self.red_defer = os.environ.get('PQ_REDUCE_BATCH', '0') == '1'
        red_items = []                       # (index of the op that leaves the slabs, pq_reduce_item)
        red_late = {}                        # wg_ops index -> filled in once the op has its scratch
        red_slots = []                       # indices of the place-holder slots, in op order
        self._red_dummy = self._new(4)

        def red_slot():
            red_slots.append(self._op(self.bwd, abi.OP_FILL, abi.Fill(self._red_dummy.data_ptr(), 0.0, 0)))

        def red_item(fn, args, at):
            it = abi.ReduceItem()
            check(fn(C.byref(args), C.byref(it)), 'reduce item')
            red_items.append((at, it))

        def bn_bwd(out: TensorRef):
            b = out.bn
            bm = b.module
            sc, tk = self._bn_scratch(b.C)
            a = abi.BnBwdFin(ptr(b.bwd_partial), ptr(bm.weight), ptr(b.invstd), flat.grad_ptr(bm.weight), flat.grad_ptr(bm.bias),
                             ptr(b.p), ptr(b.q), ptr(b.r), b.bwd_nblk, b.ldp, b.C, int(self.bn_train), b.count, sc, tk)
            self._op(self.bwd, abi.OP_BN_BWD_FIN, a)

        def zero_params(m):
            for p in m.parameters():
                o = flat.index[id(p)]
                self._op(self.bwd, abi.OP_FILL, abi.Fill(flat.grad_views[o].data_ptr(), 0.0, p.numel()))

        self._loss_fused_at, self._loss_extra = None, []
        fuse_loss = 2 <= len(heads) <= 3 and os.environ.get('PQ_LOSS_FUSE', '1') != '0'
        # slot 0: one batched transpose of the 1x1 weights whose data gradient is needed (filled in after the loop)
        tr_idx = self._op(self.bwd, abi.OP_TRANSPOSE, abi.Transpose(None, 0, 0))
        tr_items = []
        # slot 1: clears the accumulators of the backward statistics hand-over when the forward cannot do it (_handover_bn_bwd)
        self._bwd_bfill_slot = self._op(self.bwd, abi.OP_FILL, abi.Fill(self._new(4).data_ptr(), 0.0, 0))

        def transposed(conv, wsrc):
            """Device address of the [I][O] copy of a 1x1 weight, refreshed at the start of every backward."""
            O_, I_ = conv.out_channels, conv.in_channels
            wt = self._new(I_ * ((O_ + 3) // 4 * 4), fill=0.0)      # rows padded to 16 bytes, pad columns stay zero
            tile0 = tr_items[-1][4] + tr_items[-1][6] if tr_items else 0
            tiles_c = (I_ + 31) // 32
            tr_items.append((wsrc, wt.data_ptr(), O_, I_, tile0, tiles_c, ((O_ + 31) // 32) * tiles_c))
            return wt.data_ptr()

        for i in range(len(mods) - 1, -1, -1):
            m = mods[i]
            t = m._type
            lo_mark = None
            if t == 'yolo':
                slot, hm, a = head_at[i]
                la, _, stride, _ = self._head_loss_args[slot]
                a.received += 1
                if self._loss_fused_at is None or not fuse_loss:
                    idx = self._op(self.bwd, abi.OP_LOSS_BWD, la)
                    self.loss_ops_b.append((idx, slot, stride))
                    if fuse_loss:
                        self._loss_fused_at = idx
                else:
                    # the gradients of this head come out of the FIRST head's launch (pq_yolo_loss_args.also): the per-head kernels
                    # are latency bound, ~26 us each on the critical chain whatever the head size
                    self._loss_extra.append(la)
                    self.loss_ops_b.append((-len(self._loss_extra), slot, stride))
            elif t == 'convolutional' and i in self.ir_expand:
                pass             # fused expand layer: every gradient of it was written by the pq_xdw_bwd of its depthwise layer
            elif t == 'convolutional' and i in self.ir_fused and T[i].grad is not None and T[i].n_consumers > 0:
                # fused depthwise + expand backward (irbwd.hip): walking kernel + fold + fix-up, main stream
                out, mid = T[i], T[self.ir_fused[i]]
                ie = self.ir_fused[i]
                conv, conv_e = m.conv, mods[ie].conv
                x_in = mid.ir_src
                if out.bn.bwd_partial is None:
                    raise RuntimeError('internal: BN layer %d never finalised its gradient' % i)
                bn_bwd(out)
                be = mid.bn
                M_ = x_in.rows
                need = l.pq_xdw_bwd_scratch_bytes(B, x_in.H, x_in.W, x_in.C, out.C, conv.stride[0])
                buf = self._new(need // 4 + 64)
                a = abi.XdwBwd()
                a.inp, a.we = x_in.src(), ptr(conv_e.weight)
                a.mid_scale, a.mid_shift, a.mid_mean, a.mid_invstd, a.mid_gamma = ptr(be.scale), ptr(be.shift), ptr(be.mean), ptr(be.invstd), \
                    ptr(be.module.weight)
                a.mid_act = mid.act
                a.dy, a.wd = out.gsrc(self.bn_train), ptr(conv.weight)
                a.dx = self._sink(x_in, l.pq_xdw_bwd_fix_blocks(M_, x_in.C))
                a.dwe, a.dwd = flat.grad_ptr(conv_e.weight), flat.grad_ptr(conv.weight)
                a.dgamma, a.dbeta = flat.grad_ptr(be.module.weight), flat.grad_ptr(be.module.bias)
                a.moments, a.scratch = mid.ir_moments.data_ptr(), ptr(buf)
                a.B, a.IH, a.IW, a.OH, a.OW, a.I, a.E, a.stride = B, x_in.H, x_in.W, out.H, out.W, x_in.C, out.C, conv.stride[0]
                self._op(self.bwd, abi.OP_XDW_BWD, a)
                lo = min(flat.offsets[flat.index[id(p_)]] for p_ in list(m.parameters()) + list(mods[ie].parameters()))
                self.segments.append((len(self.bwd), lo))
            elif t == 'convolutional':
                out = T[i]
                conv = m.conv
                wsrc = ptr(self._wq[i][0]) if self.qat else ptr(conv.weight)          # what the forward convolved with
                wdst = ptr(self._wq[i][2]) if self.qat else flat.grad_ptr(conv.weight)   # where its gradient goes
                if out.n_consumers == 0 or out.grad is None:
                    zero_params(m)
                    if self.qat:
                        self._op(self.bwd, abi.OP_FILL, abi.Fill(wdst, 0.0, conv.weight.numel()))
                else:
                    if self.qat:                    # through the activation observer: out <- the un-quantised producer tensor
                        inner = self._inner[i]
                        if getattr(out, 'fuse_inner', None) is None:
                            self._qact_bwd(out, inner, ('a', i))
                        out = inner                 # (fused: the consumers' sinks already finalised inner.grad)
                    if out.bn is not None:
                        if out.bn.bwd_partial is None:
                            raise RuntimeError('internal: BN layer %d never finalised its gradient' % i)
                        # frozen statistics (model.eval() or tools.freeze_bn_stats): p = gamma*invstd, q = r = 0
                        bn_bwd(out)
                    dy = out.gsrc(self.bn_train)
                    k, s = conv.kernel_size[0], conv.stride[0]
                    if i == 0:
                        red_slot()              # in front of the step's last kernel, not behind it
                        need = l.pq_stem3x3_wgrad_scratch_bytes(B, out.H, out.W, out.C, 3)
                        a = abi.StemWgrad(dy, ptr(self.qimg) if self.qat else None, wdst, None, B, 3, self.IH, self.IW, out.H,
                                          out.W, out.C, s)
                        idx = self._op(self.bwd, abi.OP_STEM_WGRAD, a)
                        if not self.qat:
                            self.patch_x.append(('bwd', idx, 'stw'))
                        wg_ops.append((idx, 'stw', need))
                    else:
                        src = T[resolve(i - 1, i)]
                        fused_blocks = 0
                        if k == 1 and s == 1 and src.needs_grad and conv.bias is None:
                            # narrow layer over many rows: input and weight gradient from one pass over dy (pwbwd.hip)
                            if l.pq_conv1x1_bwd_pays(out.rows, src.C, out.C):
                                fused_blocks = l.pq_conv1x1_bwd_blocks(out.rows, src.C, out.C)
                        if fused_blocks:
                            need = l.pq_conv1x1_bwd_scratch_bytes(out.rows, src.C, out.C)
                            buf = self._new(max(need, 64) // 4 + 16)
                            a = abi.Conv1x1Bwd(dy, wsrc, src.src(), self._sink(src, fused_blocks), wdst, ptr(buf),
                                               out.rows, src.C, out.C)
                            if self.red_defer:
                                red_item(l.pq_conv1x1_bwd_reduce_item, a, len(self.bwd))
                                a.dw = None
                            self._op(self.bwd, abi.OP_CONV1X1_BWD, a)
                        elif k == 1:
                            need = l.pq_conv1x1_wgrad_scratch_bytes(out.rows, src.C, out.C)
                            a = abi.Conv1x1Wgrad(dy, src.src(), wdst,
                                                 flat.grad_ptr(conv.bias) if conv.bias is not None else None, None,
                                                 out.rows, src.C, out.C, s, out.H, out.W, src.H, src.W)
                            idx = self._op(self.bwd, abi.OP_CONV1X1_WGRAD, a)
                            wg_ops.append((idx, 'c1w', need))
                            if src.needs_grad and s == 1:
                                d = abi.Conv1x1Dgrad(dy, wsrc, self._sink(src, l.pq_conv1x1_blocks_k(src.rows, src.C, out.C, 1)), out.rows,
                                                     src.C, out.C, 1, 0, 0, 0, 0, transposed(conv, wsrc))
                                self._op(self.bwd, abi.OP_CONV1X1_DGRAD, d)
                            elif src.needs_grad:
                                # strided 1x1: dz only lands on every s-th pixel.  The GEMM accumulates into a (zeroed if it
                                # is the first contribution) gradient plane; if it was the last contribution an in-place
                                # element-wise pass finalises the whole plane (activation derivative + BN sums).
                                sk = self._sink(src, l.pq_ew_blocks(src.rows, src.C), addend_ok=False)
                                if not sk.accumulate:
                                    self._op(self.bwd, abi.OP_FILL, abi.Fill(ptr(src.grad), 0.0, src.rows * src.ld))
                                acc = abi.GSink(sk.g, sk.y, None, None, None, None, None, src.ld, src.ld, 0, 1, 0, 0)
                                d = abi.Conv1x1Dgrad(dy, wsrc, acc, out.rows, src.C, out.C, s, out.H, out.W, src.H,
                                                     src.W, transposed(conv, wsrc))
                                self._op(self.bwd, abi.OP_CONV1X1_DGRAD, d)
                                if sk.finalize:
                                    sk.accumulate = 0
                                    r = abi.RouteBwd(ptr(src.grad), sk, src.B, src.H, src.W, src.C, src.ld, 0, 1)
                                    self._op(self.bwd, abi.OP_ROUTE_BWD, r)
                        elif conv.groups != conv.in_channels or conv.in_channels // conv.groups > 1:
                            gw = conv.in_channels // conv.groups
                            need = l.pq_gconv3x3_wgrad_scratch_bytes(B, out.H, out.W, out.C, gw)
                            a = abi.GconvWgrad(dy, src.src(), wdst, None, B, src.H, src.W, out.H, out.W,
                                               out.C, gw, s)
                            idx = self._op(self.bwd, abi.OP_GCONV_WGRAD, a)
                            wg_ops.append((idx, 'gcw', need))
                            if src.needs_grad:
                                d = abi.GconvDgrad(dy, wsrc, self._sink(src, l.pq_gconv3x3_blocks(B, src.H, src.W)),
                                                   B, src.H, src.W, out.H, out.W, out.C, gw, s)
                                self._op(self.bwd, abi.OP_GCONV_DGRAD, d)
                        else:
                            # (a folded shortcut's output is finalised through its conv branch's y: the walking kernel reads ONE tensor as
                            # the conv's input and the sink's y, so such an input takes the separate data / weight gradient kernels)
                            walk = src.needs_grad and getattr(src, 'fold_sa', None) is None and \
                                (s == 1 or (src.H % 2 == 0 and src.W % 2 == 0))
                            if walk:
                                nblk = l.pq_dw3x3_bwd_walk_blocks(B, src.H, src.W, out.C, s)
                                sink, sin = self._sink(src, nblk), src.src()
                                if not (sink.y == sin.x and sink.ldy == sin.ld and sink.scale == sin.scale and
                                        sink.shift == sin.shift and sink.act == sin.act):
                                    raise RuntimeError('internal: layer %d: the depthwise conv\'s gradient sink is not its input tensor' % i)
                            if walk:
                                # the WALKING backward (dwwalk.hip): both gradients from one pass over (g, y_out, y_in) on the main
                                # stream -- 4 tensor passes instead of the 7 of a data- and a weight-gradient kernel -- with one
                                # statistics / weight-gradient row per workgroup; only the reduction of those few rows goes to the
                                # weight-gradient stream.  (Separate data / weight gradient kernels remain for stride-2 planes of odd size
                                # and for a conv whose input needs no gradient.)
                                buf = self._new(nblk * out.C * 9 + 16)
                                a = abi.DwBwd(dy, sin, wsrc, sink, None, ptr(buf), B, src.H, src.W, out.H, out.W, out.C, s, 5)
                                self._op(self.bwd, abi.OP_DW_BWD, a)
                                a = abi.DwBwd(abi.GSrc(), abi.Src(), None, abi.GSink(), wdst, ptr(buf), B, src.H, src.W, out.H,
                                              out.W, out.C, s, 6)
                                if self.red_defer:
                                    red_item(l.pq_dw3x3_bwd_reduce_item, a, len(self.bwd) - 1)
                                else:
                                    idx = self._op(self.bwd, abi.OP_DW_BWD, a)
                                    # the reduction of a few rows is not urgent: it rides with the next weight gradient's fork
                                    self.bwd[idx].flags = abi.OPF_SIDE | abi.OPF_LAZY
                            else:
                                need = l.pq_dw3x3_wgrad_scratch_bytes(B, out.H, out.W, out.C)
                                a = abi.DwWgrad(dy, src.src(), wdst, None, B, src.H, src.W, out.H, out.W,
                                                out.C, s)
                                idx = self._op(self.bwd, abi.OP_DW_WGRAD, a)
                                wg_ops.append((idx, 'dww', need))
                                d = abi.DwDgrad(dy, wsrc, self._sink(src, l.pq_ew_blocks(src.rows, src.C)), B, src.H,
                                                src.W, out.H, out.W, out.C, s)
                                self._op(self.bwd, abi.OP_DW_DGRAD, d)
                if i != 0:
                    red_slot()
                plist = list(m.parameters())
                if plist:
                    lo = min(flat.offsets[flat.index[id(p)]] for p in plist)
                    self.segments.append((len(self.bwd), lo))
            elif t == 'shortcut':
                out = T[i]
                if self.qat and out.grad is not None:
                    self._qact_bwd(out, self._inner[i], ('a', i))
                    out = self._inner[i]
                if out.grad is not None and getattr(out, 'fold_sa', None) is not None:
                    sa, sb = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                    if sa.grad is None or sa.received != sa.n_consumers:
                        raise RuntimeError('internal: folded shortcut %d: the gradient of its conv branch was never finalised' % i)
                    if sb.grad is None and sb.received == 0 and sb.n_consumers > 1 and getattr(sb, 'addend', None) is None:
                        sb.addend = out.grad             # read by b's next sink (pq_gsink.addend): no launch here
                        sb.received += 1
                    else:
                        r = abi.RouteBwd(ptr(out.grad), self._sink(sb, l.pq_ew_blocks(sb.rows, sb.C)), sb.B, sb.H, sb.W, sb.C, out.ld, 0, 1)
                        self._op(self.bwd, abi.OP_ROUTE_BWD, r)
                elif out.grad is not None and getattr(out, 'alias_sa', None) is not None:
                    sa, sb = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                    if sa.grad is not None or sa.received:
                        raise RuntimeError('internal: aliased shortcut %d: its conv branch already has a gradient' % i)
                    sa.grad, sa.received = out.grad, sa.n_consumers
                    b = sa.bn
                    b.bwd_nblk = l.pq_ew_blocks(sa.rows, sa.C)
                    b.bwd_partial = self._new(b.bwd_nblk * 2 * b.ldp)
                    sk = abi.GSink(ptr(out.grad), ptr(sa.raw), ptr(b.scale), ptr(b.shift), ptr(b.mean), ptr(b.invstd), ptr(b.bwd_partial),
                                   out.ld, sa.ld, sa.act, 0, 3, b.ldp)
                    self._op(self.bwd, abi.OP_ROUTE_BWD, abi.RouteBwd(ptr(out.grad), sk, sa.B, sa.H, sa.W, sa.C, out.ld, 0, 1))
                    if sb.grad is None and sb.received == 0 and sb.n_consumers > 1 and getattr(sb, 'addend', None) is None:
                        sb.addend = out.grad
                        sb.received += 1
                    else:
                        r = abi.RouteBwd(ptr(out.grad), self._sink(sb, l.pq_ew_blocks(sb.rows, sb.C)), sb.B, sb.H, sb.W, sb.C, out.ld, 0, 1)
                        self._op(self.bwd, abi.OP_ROUTE_BWD, r)
                elif out.grad is not None:
                    sa, sb = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                    if sa is not sb and sa.C % 4 == 0 and not self.qat:
                        # both addends receive the sum's gradient: ONE pass over it feeds the two sinks (pq_route_bwd dx2)
                        k1 = self._sink(sa, l.pq_ew_blocks(sa.rows, sa.C))
                        k2 = self._sink(sb, l.pq_ew_blocks(sb.rows, sb.C))
                        r = abi.RouteBwd(ptr(out.grad), k1, sa.B, sa.H, sa.W, sa.C, out.ld, 0, 1, None, 0, 0, 0.0, k2)
                        self._op(self.bwd, abi.OP_ROUTE_BWD, r)
                    else:
                        for srcT in (sa, sb):
                            r = abi.RouteBwd(ptr(out.grad), self._sink(srcT, l.pq_ew_blocks(srcT.rows, srcT.C)), srcT.B, srcT.H,
                                             srcT.W, srcT.C, out.ld, 0, 1)
                            self._op(self.bwd, abi.OP_ROUTE_BWD, r)
            elif t == 'route':
                if not m.single and len(m._layers) > 1:
                    out = T[i]
                    if self.qat and out.grad is not None:
                        self._qact_bwd(out, self._inner[i], ('a', i))
                        out = self._inner[i]
                    off = 0
                    for j in m._layers:
                        srcT = T[resolve(j, i)]
                        if out.grad is not None:
                            r = abi.RouteBwd(ptr(out.grad), self._sink(srcT, l.pq_ew_blocks(srcT.rows, srcT.C)), srcT.B,
                                             srcT.H, srcT.W, srcT.C, out.ld, off, 1)
                            self._op(self.bwd, abi.OP_ROUTE_BWD, r)
                        off += srcT.C
            elif t == 'avgpool':
                out = T[i]
                srcT = T[resolve(i - 1, i)]
                if out.grad is not None:
                    r = abi.RouteBwd(ptr(out.grad), self._sink(srcT, l.pq_ew_blocks(srcT.rows, srcT.C)), srcT.B, srcT.H, srcT.W,
                                     srcT.C, out.ld, 0, 1, None, 0, 1, 1.0 / (srcT.H * srcT.W))
                    self._op(self.bwd, abi.OP_ROUTE_BWD, r)
            elif t == 'scale_channels':
                out = T[i]
                gate, feat = T[resolve(i - 1, i)], T[resolve(m._from, i)]
                if out.grad is not None:
                    # gradient of the gate's pre-sigmoid value: s(1-s) * sum_p gout * feature
                    if gate.grad is None:
                        gate.grad = self._new(gate.rows, gate.ld, fill=0.0)
                    gate.received += 1
                    pr = abi.PoolReduce(feat.src(), ptr(out.grad), out.ld, ptr(gate.raw), gate.ld, ptr(gate.grad), gate.ld,
                                        feat.B, feat.H * feat.W, feat.C, 1.0)
                    self._op(self.bwd, abi.OP_POOL_REDUCE, pr)
                    r = abi.RouteBwd(ptr(out.grad), self._sink(feat, l.pq_ew_blocks(feat.rows, feat.C)), feat.B, feat.H, feat.W,
                                     feat.C, out.ld, 0, 1, ptr(gate.raw), gate.ld, 0, 1.0)
                    self._op(self.bwd, abi.OP_ROUTE_BWD, r)
            elif t == 'upsample':
                out = T[i]
                srcT = T[resolve(i - 1, i)]
                f = int(m.scale_factor)
                if out.grad is not None:
                    r = abi.RouteBwd(ptr(out.grad), self._sink(srcT, l.pq_ew_blocks(srcT.rows, srcT.C)), srcT.B, srcT.H,
                                     srcT.W, srcT.C, out.ld, 0, f)
                    self._op(self.bwd, abi.OP_ROUTE_BWD, r)
        if tr_items:
            table = (abi.TransposeItem * len(tr_items))(*[abi.TransposeItem(*it[:6]) for it in tr_items])
            dev_table = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
            self.keep.append(dev_table)
            self.bwd[tr_idx].u.tr = abi.Transpose(dev_table.data_ptr(), len(tr_items), tr_items[-1][4] + tr_items[-1][6])
        # Two-stream forward, plain weights: the transposed copies only depend on the parameters, so the launch moves off the
        # critical chain (it was the first kernel of every backward, ~15 us with nothing beside it) into the forward
        # list (slot in front of the loss ops) as a side-stream op without a fork: it runs behind the head branches while the
        # main stream finishes the neck, and the join in front of the last head's finalize covers it.
        # (QAT transposes the fake-quantised weights of the step, which are main-stream products: stays in the backward.)
        # NOFORK is only safe behind a side op of THIS forward that did fork (the head-branch layers): the side stream is then
        # past the previous step's optimiser update / backward, which a fork-less op at its head would not be.
        self.fwd_side_slots_ok = getattr(self, 'fwd_two_streams', False) and bool(getattr(self, 'fwd_side_layers', []))
        self.tr_in_forward = bool(tr_items) and self.fwd_side_slots_ok and not self.qat
        if self.tr_in_forward:
            op = abi.Op()
            op.code = abi.OP_TRANSPOSE
            op.u.tr = self.bwd[tr_idx].u.tr
            op.flags = abi.OPF_SIDE | abi.OPF_NOFORK
            self.fwd[self._fwd_tr_slot] = op
        if not tr_items or self.tr_in_forward:      # nothing (left) to transpose here: turn slot 0 into a no-op
            dummy = self._new(4)
            nop = abi.Op()
            nop.code = abi.OP_FILL
            nop.u.fill = abi.Fill(dummy.data_ptr(), 0.0, 0)
            self.bwd[tr_idx] = nop
        # weight-gradient ops only feed the optimiser: they get private scratch and may run on the side stream
        late = {'c1w': l.pq_conv1x1_wgrad_reduce_item, 'gcw': l.pq_gconv3x3_wgrad_reduce_item}
        for idx, field, need in wg_ops:
            buf = self._new(max(need, 64) // 4 + 16)
            a = getattr(self.bwd[idx].u, field)
            a.scratch = buf.data_ptr()
            self.bwd[idx].flags = abi.OPF_SIDE
            if self.red_defer and field in late:
                red_item(late[field], a, idx)
                a.dw = None
        self._emit_reduce_batches(red_items, red_slots)
        self._handover_bn_bwd()

    def _emit_reduce_batches(self, items, slots):
        """Turn some of the place-holder slots into pq_reduce_batch launches (side stream): every deferred item goes into the first
        real slot behind the op that leaves its slabs; a slot becomes real when `per` items are waiting (PQ_REDUCE_GROUPS launches
        per step, default 6) and the last slot with anything waiting always does.  Gradient-ready marks (self.segments) in front of
        which a slab is still unreduced are dropped: a DDP bucket may only close behind the launch that covers its layers."""
        self.red_batches = 0
        if not items:
            return
        l = lib()
        groups = max(1, int(os.environ.get('PQ_REDUCE_GROUPS', '6')))
        items = all_items = sorted(items, key=lambda e: e[0])
        # the trailing items whose slabs are MAIN-stream products (walking depthwise backward, fused narrow 1x1 backward of the first
        # layers) form the last launch: PQ_OPF_LAZY, i.e. at the tail of a joined backward it runs on the main stream BESIDE the
        # stem's weight gradient (pq_run_ops2), not in front of it on the weight-gradient stream
        n_head = len(items)
        while n_head > 0 and not (self.bwd[items[n_head - 1][0]].flags & abi.OPF_SIDE):
            n_head -= 1
        if os.environ.get('PQ_REDUCE_LAZY_TAIL', '1') == '0':
            n_head = len(items)
        lazy_tail = [it for _, it in items[n_head:]]
        items = items[:n_head]
        per = max(1, (len(items) + groups - 1) // groups)
        real, k, waiting, done_at = [], 0, [], []           # done_at: (op that leaves the slabs, slot of the launch that sums them)
        for si, slot in enumerate(slots):
            while k < len(items) and items[k][0] < slot:
                waiting.append(items[k][1])
                k += 1
            if waiting and (len(waiting) >= per or si == len(slots) - 1 or k == len(items)):
                real.append((slot, waiting, 0))
                waiting = []
                done_at += [(e[0], slot) for e in items[len(done_at):k]]
        done_at += [(e[0], slots[-1]) for e in all_items[n_head:]]
        if lazy_tail:
            if real and real[-1][0] == slots[-1]:
                real[-1] = (slots[-1], real[-1][1] + lazy_tail, 0)
            else:
                real.append((slots[-1], lazy_tail, abi.OPF_LAZY))
        if k < len(items) or waiting:
            raise RuntimeError('internal: a deferred weight-gradient reduction has no slot behind it')
        for slot, its, extra in real:
            table = (abi.ReduceItem * len(its))(*its)
            nblk = l.pq_reduce_batch_plan(table, len(its))
            if nblk <= 0:
                raise RuntimeError('pq_reduce_batch_plan: ' + l.pq_last_error().decode())
            dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
            self.keep.append(dev)
            op = abi.Op()
            op.code = abi.OP_REDUCE_BATCH
            op.u.red = abi.ReduceBatch(dev.data_ptr(), len(its), nblk)
            op.flags = abi.OPF_SIDE | extra
            self.bwd[slot] = op
        self.red_batches = len(real)
        if len(done_at) != len(all_items):
            raise RuntimeError('internal: deferred reductions lost')
        # a mark is valid when no op in front of it has left slabs whose launch comes at or behind it
        self.segments = [sg for sg in self.segments if not any(at < sg[0] <= slot for at, slot in done_at)]

    def _handover_bn_bwd(self):
        """Backward twin of _handover_bn_stats: drop the pq_bn_bwd_finalize launch between the kernel whose sink finalises a
        BatchNorm output's gradient (the PRODUCER: data gradient of the next 1x1 conv, plain or fused; walking depthwise backward;
        shortcut / route / upsample backward) and the kernels that read that gradient operand (the CONSUMERS: the layer's own data
        and weight gradient -- 1x1 on the ring kernels, fused narrow 1x1, walking depthwise).  The producer adds its sums into exact
        fixed-point accumulators (pq_gsink.stat_acc), every consumer derives p / q / r in its prologue (its dy_stat) and the ones
        with a gradient sink of their own store dgamma / dbeta and the vectors.  A finalize launch costs its own 4-9 us on the
        critical chain plus a kernel boundary; the weight gradient's fork then hangs off the producer.  The accumulators are
        cleared by a side-stream op at the end of the forward (two-stream forward), else by the first op of the backward."""
        self.bn_bwd_handover = 0
        self.bn_bwd_handover_producers = set()          # op codes of the kernels that add (tests)
        self.bn_bwd_handover_consumers = set()
        # Scope (PQ_BN_BWD_ATOMIC): 'dw' (default) = the layers whose only reader is the walking depthwise backward, one workgroup
        # column per channel chunk; 'all' = every layer whose producer and readers support it (61 of 81 on mobilenetv2-fpn);
        # '0' = none.  MEASURED (round 4, mobilenetv2-fpn 512 x 512 batch 16, same box): none 9.81-9.84 ms, 'dw' 9.76-9.78,
        # 'all' 10.06-10.08, 'all' with at most 16 / 32 / 64 / 128 / 256 row blocks 9.76 / 9.81 / 9.81 / 9.85 / 9.86: a GEMM reader
        # re-derives the whole contraction's coefficients in EVERY workgroup (K x 32 B x slots from L2 per workgroup, thousands of
        # workgroups) and the producers' same-address atomics serialise, which costs more than the 6.5 us launch it replaces.
        scope = os.environ.get('PQ_BN_BWD_ATOMIC', 'dw')
        if not self.bn_train or self.qat or scope == '0':
            return
        l = lib()
        maxblk = 640                                                 # row blocks adding to one channel (same-address atomics serialise)
        only_dw = scope not in ('all', '1')
        sink_fields = {abi.OP_CONV1X1_DGRAD: ('dx',), abi.OP_CONV1X1_BWD: ('dx',), abi.OP_DW_BWD: ('dx',), abi.OP_ROUTE_BWD: ('dx', 'dx2')}
        dy_ops = (abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_WGRAD, abi.OP_CONV1X1_BWD, abi.OP_DW_BWD, abi.OP_GCONV_DGRAD, abi.OP_GCONV_WGRAD,
                  abi.OP_DW_DGRAD, abi.OP_DW_WGRAD, abi.OP_STEM_WGRAD)
        writers = (abi.OP_CONV1X1_DGRAD, abi.OP_CONV1X1_BWD, abi.OP_DW_BWD)
        picks, total = [], 0
        for k in range(1, len(self.bwd) - 1):
            op = self.bwd[k]
            if op.code != abi.OP_BN_BWD_FIN or op.flags:
                continue
            f = op.u.bnb
            if not f.train or f.nblk > maxblk or not f.partial:
                continue
            # ---- the producer: the op whose sink finalises this tensor's gradient
            prod = None
            for j in range(k - 1, -1, -1):
                pj = self.bwd[j]
                if pj.code not in sink_fields:
                    continue
                pa = getattr(pj.u, abi.OP_FIELD[pj.code])
                for name in sink_fields[pj.code]:
                    sk = getattr(pa, name)
                    if sk.g and sk.partial == f.partial and sk.finalize:
                        prod = (pj, pa, sk)
                if prod:
                    break
            if not prod:
                continue
            pj, pa, sk = prod
            if pj.flags or sk.stat_acc or sk.ldp != f.ldp or not l.pq_bn_bstat_ok(pj.code, C.byref(pa), 0):
                continue
            # ---- the consumers: every op that reads the gradient operand through these p / q / r vectors
            cons, ok = [], True
            for j in range(k + 1, len(self.bwd)):
                cj = self.bwd[j]
                if cj.code not in dy_ops:
                    continue
                ca = getattr(cj.u, abi.OP_FIELD[cj.code])
                if not ca.dy.p or ca.dy.p != f.p:
                    continue
                if not hasattr(ca, 'dy_stat') or ca.dy_stat.acc or not l.pq_bn_bstat_ok(cj.code, C.byref(ca), 1):
                    ok = False
                    break
                cons.append((cj, ca))
            if not ok or not any(cj.code in writers and not cj.flags for cj, _ in cons):
                continue
            if only_dw and not all(cj.code == abi.OP_DW_BWD for cj, _ in cons):
                continue
            slots = l.pq_bn_stat_slots(f.nblk)
            nbytes = _round_up(l.pq_bn_stat_bytes(f.ldp, slots), 64)
            picks.append((k, sk, cons, f, slots, total, total + nbytes))
            total += nbytes + 64
            self.bn_bwd_handover_producers.add(pj.code)
            self.bn_bwd_handover_consumers.update(cj.code for cj, _ in cons)
        if not picks:
            return
        arena = torch.zeros(total // 8, device=self.device, dtype=torch.int64)
        self.keep.append(arena)
        base = arena.data_ptr()
        nop = abi.Op()
        nop.code = abi.OP_FILL
        nop.u.fill = abi.Fill(base, 0.0, 0)
        for k, sk, cons, f, slots, off_acc, off_bad in picks:
            sk.stat_acc, sk.stat_bad, sk.stat_slots = base + off_acc, base + off_bad, slots
            for _, ca in cons:
                ca.dy_stat = abi.BnBstat(base + off_acc, base + off_bad, f.gamma, f.invstd, f.dgamma, f.dbeta, f.p, f.q, f.r,
                                         slots, f.ldp, f.count, 1)
            self.bwd[k] = nop
        fill = abi.Fill(base, 0.0, total // 4)
        if self.fwd_side_slots_ok:
            # behind the head branches on the forward's side stream (ordered after this forward's first fork, i.e. after the
            # previous backward; covered by the join in front of the last head's finalize): nothing on the critical chain
            op = abi.Op()
            op.code = abi.OP_FILL
            op.u.fill = fill
            op.flags = abi.OPF_SIDE | abi.OPF_NOFORK
            self.fwd[self._fwd_bfill_slot] = op
        else:
            self.bwd[self._bwd_bfill_slot].u.fill = fill
        self.bn_bwd_handover = len(picks)

    # ------------------------------------------------------------------------------------------ run
    def _patch(self, x, target):
        xp = x.data_ptr()
        for which, idx, field in self.patch_x:
            arr = self.fwd_arr if which == 'fwd' else self.bwd_arr
            if field == 'qa':
                arr[idx].u.qa.inp.x = xp          # QAT: the batch enters through the input observer
            elif field == 'lay':
                arr[idx].u.lay.inp.x = xp         # dense first conv: the patch matrix is gathered from the NCHW batch
            else:
                getattr(arr[idx].u, field).image = xp
        if target is not None:
            pick = {8: (0, 3), 16: (1, 4), 32: (2, 5)}       # reference model/interpreter.py:16-20
            for lst, arr in ((self.loss_ops_f, self.fwd_arr), (self.loss_ops_b, self.bwd_arr)):
                for idx, slot, stride in lst:
                    li, bi = pick[stride]
                    la = arr[idx].u.loss if idx >= 0 else self.loss_bwd_extra[-idx - 1]
                    la.label = target[li].data_ptr()
                    la.boxes = target[bi].data_ptr()
                    la.M = target[bi].shape[1]

    def set_totals(self, t):
        """Where the last head's loss finalize writes the step's totals (4 + n_heads floats)."""
        self.fwd_arr[self._last_loss_fin].u.loss.totals = t.data_ptr()

    def patch_upstream(self, grads):
        """Upstream gradient scalars of the step's 4 + n_heads outputs (device tensors of one float each, or None) -> the
        backward loss ops read them in place: no host-side arithmetic, no copy."""
        ptrs = [None if g is None else g.data_ptr() for g in grads]
        for idx, slot, _ in self.loss_ops_b:
            la = self.bwd_arr[idx].u.loss if idx >= 0 else self.loss_bwd_extra[-idx - 1]
            for k in range(4):
                la.gup[k] = ptrs[k]
            la.gup[4] = ptrs[4 + slot]

    def run_forward(self, x, target):
        self._patch(x, target)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        side = self.model._side_stream() if getattr(self, 'fwd_two_streams', False) else None
        if side is not None:
            check(lib().pq_run_ops2(self.fwd_arr, len(self.fwd), st, C.c_void_p(side.cuda_stream)), 'forward step')
            return
        check(lib().pq_run_ops(self.fwd_arr, len(self.fwd), st), 'forward step')

    def run_backward(self, reducer=None):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        l = lib()
        side = self.model._side_stream()
        sd = C.c_void_p(side.cuda_stream) if side is not None else None
        if reducer is None:
            check(l.pq_run_ops2(self.bwd_arr, len(self.bwd), st, sd), 'backward step')
            if self.post_arr is not None:        # QAT: weight gradients through the weight observers (needs every wgrad done)
                check(l.pq_run_ops(self.post_arr, len(self.post_bwd), st), 'backward step (weight observers)')
            return
        if self.post_arr is not None:
            raise NotImplementedError('quantisation-aware training is single-GPU (as in the reference, trainer.py:148)')
        # bucketed replay: launch the ops up to each mark WITHOUT joining the two compute streams (the weight gradients keep
        # overlapping the data-gradient chain across bucket boundaries); the finished range is handed to the reducer, whose
        # communication stream waits for BOTH compute streams at that point.  One join at the very end.
        flat = self.model._flat
        start, hi = 0, flat.grad.numel()
        marks = reducer.pick_marks(self.segments, hi)
        base = C.addressof(self.bwd_arr)
        for op_end, lo in marks:
            n = op_end - start
            if n > 0:
                sub = C.cast(C.c_void_p(base + start * C.sizeof(abi.Op)), C.POINTER(abi.Op))
                check(l.pq_run_ops2_nojoin(sub, n, st, sd), 'backward step')
            reducer.reduce_range(flat.grad, lo, hi, also_wait=side)
            start, hi = op_end, lo
        if start < len(self.bwd):
            sub = C.cast(C.c_void_p(base + start * C.sizeof(abi.Op)), C.POINTER(abi.Op))
            check(l.pq_run_ops2_nojoin(sub, len(self.bwd) - start, st, sd), 'backward step')
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)          # the only join of the two compute streams
        if hi > 0:
            reducer.reduce_range(flat.grad, 0, hi)
        reducer.finish()
