"""``DetectionModel`` -- same constructor, attributes, ``state_dict`` layout and ``forward(x, target=None)`` contract as
the reference ``model/interpreter.py:22-88``; the layer loop itself is compiled into HIP kernel launches by
:mod:`pqdet_amd.engine` instead of being interpreted module by module.

* eval  : ``forward(x)`` -> ``Tensor[B, sum(H*W*A), 5+C]`` (heads concatenated in cfg order, reference :74-76)
* train : ``forward(x, (label_s, label_m, label_l, box_s, box_m, box_l))`` -> dict with ``loss``, ``giou_loss``,
  ``conf_loss``, ``class_loss`` (each ``(1,)``) and ``loss_per_branch`` (list of ``(1,)`` per head, cfg order),
  reference :77-85; targets are routed to heads by stride through ``_TARGET_MAP`` (:16-20).
  ``out['loss'].mean().backward()`` runs the compiled backward and leaves ``param.grad`` as views into one flat buffer.
"""
from typing import IO, Union

import torch
from torch import nn

from .. import abi
from ..engine import FlatParams, Plan
from .parser import Parser


def item_getter(*items):
    def getter(x):
        if x is None:
            return None
        return tuple(x[i] for i in items)
    return getter


_TARGET_MAP = {
    8: item_getter(0, 3),
    16: item_getter(1, 4),
    32: item_getter(2, 5),
}


class _StepFn(torch.autograd.Function):
    """Whole-network forward/backward as one autograd node (the anchor parameter only ties it into the graph)."""

    """Outputs: 4 + n_heads tensors of shape (1,): loss, box, conf, cls summed over the heads (head order, like the reference's
    ``map(sum, zip(*outputs))``, model/interpreter.py:77), then every head's own total.  They are summed by the last head's loss
    kernel; the upstream gradients are read in place by the backward loss kernels -- the step adds one device copy (the
    returned values must not alias a buffer the next step overwrites) and nothing else to what the driver itself does."""

    @staticmethod
    def forward(ctx, anchor, model, plan, x, target):
        # the last head's finalize writes the totals straight into this step's own tensor (a clone of a plan buffer was one
        # more copy kernel on the chain between forward and backward)
        res = torch.empty_like(plan.loss_total)
        plan.set_totals(res)
        plan.run_forward(x, target)
        plan.set_totals(plan.loss_total)        # launches carry their arguments: leave no pointer to `res` behind
        ctx.model, ctx.plan = model, plan
        ctx.x, ctx.target = x, target           # keep the batch alive until backward has consumed it
        ctx.set_materialize_grads(False)
        return tuple(res[k:k + 1] for k in range(res.numel()))

    @staticmethod
    def backward(ctx, *grads):
        plan, model = ctx.plan, ctx.model
        keep = []
        for g in grads:
            if g is not None and (g.dtype != torch.float32 or g.numel() != 1 or not g.is_cuda):
                g = g.to(device=plan.device, dtype=torch.float32).reshape(-1)[:1].contiguous()
            keep.append(g)
        plan._patch(ctx.x, ctx.target)
        plan.patch_upstream(keep)
        plan.run_backward(model._reducer)
        plan.patch_upstream(plan.default_upstream)     # the gradient tensors die with this call: leave no pointer to them behind
        model._flat.assign_grads()
        return None, None, None, None, None


class AnyModel(nn.Module):

    def __init__(self, cfg: Union[str, IO], quant: bool = False, onnx: bool = False):
        super().__init__()
        self.quant = quant            # quant=True: every activation is ReLU (reference parser :408-409); QAT itself is
        self._qat = False             # switched on by tools.prepare_qat(model), like in the reference
        self.qat_observer_enabled = True
        self.qat_fake_quant_enabled = True
        self.bn_frozen = False        # tools.freeze_bn_stats: BatchNorm uses (and keeps) its running statistics while training
        if isinstance(cfg, str):
            cfg = open(cfg, 'r')
        self.module_list = nn.ModuleList(Parser(cfg).torch_layers(quant, onnx))
        cfg.close()
        self._flat = None
        self._plans = {}               # (shape, mode) -> compiled Plan, in least-recently-used order
        self._buffer_ptrs = None
        self.max_plans = int(__import__('os').environ.get('PQ_MAX_PLANS', '12'))
        self._reducer = None          # set by pqdet_amd.parallel.DistributedModel
        self._nan_event = None
        self._nan_host = None
        self._nan_dev = None
        self._nan_flags = []          # pinned host flags of the plans stepped since the last synchronous check
        self.nan_check = 'deferred'   # 'sync' | 'deferred' | 'off'
        self.overlap_wgrad = True     # run weight-gradient kernels on a side HIP stream during backward
        self._side = None

    def _side_stream(self):
        if not self.overlap_wgrad:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    # ---- flat parameter storage ------------------------------------------------------------------------
    def _ensure_flat(self, device):
        f = self._flat
        if f is None or f.flat.device != device or not f.still_valid():
            self._flat = FlatParams(self, device)
            self._plans.clear()
        return self._flat

    def flat_buffers(self):
        """(flat parameter buffer, flat gradient buffer); parameters are views into the first."""
        dev = next(self.parameters()).device
        f = self._ensure_flat(dev)
        return f.flat, f.grad

    def _plan(self, x, with_loss):
        if not x.is_cuda:
            raise abi.PQError('pqdet_amd models run on the GPU through libpqdet_hip.so; there is no CPU path '
                              '(got input on %s)' % x.device)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError('expected (B, 3, H, W) with H, W multiples of 32, got %s' % (tuple(x.shape),))
        self._ensure_flat(x.device)
        ptrs = []
        for b in self.buffers():
            if b.device != x.device:
                raise abi.PQError('model buffers live on %s but the input is on %s' % (b.device, x.device))
            ptrs.append(b.data_ptr())
        ptrs = tuple(ptrs)
        if ptrs != self._buffer_ptrs:          # a buffer was re-assigned (bn.running_mean = t): compiled plans hold its old address
            self._plans.clear()
            self._buffer_ptrs = ptrs
        bn_train = bool(self.training) and not self.bn_frozen
        key = (tuple(x.shape), bn_train, bool(with_loss), bool(self._qat), bool(self.qat_observer_enabled),
               bool(self.qat_fake_quant_enabled))
        plan = self._plans.pop(key, None)
        if plan is None:
            # a plan owns every activation / gradient buffer of its shape (nothing is shared or recomputed); multi-scale
            # training (reference: 10 input sizes) therefore keeps `max_plans` of them, least recently used first out
            while len(self._plans) >= max(1, int(self.max_plans)):
                self._plans.pop(next(iter(self._plans)))
            plan = Plan(self, tuple(x.shape), bn_train, bool(with_loss))
        self._plans[key] = plan                  # most recently used last
        return plan

    # ---- quantisation-aware training ----------------------------------------------------------------------
    def prepare_qat(self):
        """Attach observer state for the input, every conv weight / output, every shortcut sum and every concatenation
        (what ``torch.quantization.prepare_qat`` adds to the reference model): one row {min, max, scale, zero_point} each,
        kept in the buffer ``_qat_state`` (part of the state_dict) in the reference's module order."""
        index, n = {('in',): 0}, 1
        for i, m in enumerate(self.module_list):
            if m._type == 'convolutional':
                index[('w', i)], index[('a', i)] = n, n + 1
                n += 2
            elif m._type == 'shortcut' or (m._type == 'route' and not m.single and len(m._layers) > 1):
                index[('a', i)] = n
                n += 1
            elif m._type not in ('route', 'upsample', 'yolo'):
                raise NotImplementedError("layer type '%s' has no QAT support" % m._type)
        dev = next(self.parameters()).device
        state = torch.tensor([float('inf'), float('-inf'), 1.0, 0.0], device=dev).repeat(n, 1).contiguous()
        self.register_buffer('_qat_state', state)
        self._qat_index = index
        self._qat = True
        self._plans.clear()
        return self

    # A prepared QAT model presents its state under the REFERENCE's key layout (fused ``conv.bn.*`` + the observers'
    # buffers), because that is what ``trainer.py:194-201`` stores with ``'type': 'qat'`` and what ``tools.build_model``
    # (reference :236-238) loads back: checkpoints written by an unmodified trainer around this model resume, and they are
    # interchangeable with checkpoints written by the reference itself.
    def plain_state_dict(self):
        """Parameters / buffers under this module's own names (``bn.*`` + ``_qat_state``)."""
        return nn.Module.state_dict(self)

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        """Prepared QAT models: DETACHED CPU copies under the reference's checkpoint keys (``reference_state_dict``; building it
        synchronises with the device).  ``keep_vars=True`` asks for the live tensors, which only exist under this module's own
        names: it returns :meth:`plain_state_dict` (an EMA / anything that mutates the returned tensors wants that one)."""
        if not self._qat or keep_vars:
            return nn.Module.state_dict(self, *args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        from .. import tools
        if destination is None:
            from collections import OrderedDict
            destination = OrderedDict()
        for k, v in tools.qat_state_dict(self).items():
            destination[prefix + k] = v
        return destination

    def reference_state_dict(self):
        """The reference-format export of a prepared QAT model (what ``state_dict()`` returns for it), by its own name."""
        from .. import tools
        return tools.qat_state_dict(self)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if self._qat and '_qat_state' not in state_dict:
            from .. import tools
            want = set(tools.qat_state_dict(self))
            have = set(k[len('module.'):] if k.startswith('module.') else k for k in state_dict)
            missing, unexpected = sorted(want - have), sorted(have - want)
            if strict and (missing or unexpected):
                raise RuntimeError('Error(s) in loading state_dict for %s (reference QAT layout): missing keys %s, unexpected '
                                   'keys %s' % (type(self).__name__, missing[:5], unexpected[:5]))
            tools.load_qat_state_dict(self, state_dict)
            return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)
        out = nn.Module.load_state_dict(self, state_dict, strict=strict, **kw)
        self._plans.clear()
        return out

    def qat_observers(self):
        """{reference-style module name: (min, max, scale, zero_point)} for checkpoints / inspection."""
        st = self._qat_state.detach().cpu()
        out = {}
        for key, k in self._qat_index.items():
            if key[0] == 'in':
                name = 'qstub.activation_post_process'
            elif key[0] == 'w':
                name = 'module_list.%d.conv.weight_fake_quant' % key[1]
            else:
                m = self.module_list[key[1]]
                name = 'module_list.%d.%s.activation_post_process' % (key[1], 'conv' if m._type == 'convolutional' else 'ffunc')
            out[name] = tuple(st[k].tolist())
        return out

    # ---- NaN trap (reference model/loss.py:110-114), polled one step late unless nan_check == 'sync' ----
    def check_nan(self, wait: bool = True):
        if self._nan_flags:                      # sticky flags the loss kernels set in pinned host memory: nothing was queued
            if wait:
                torch.cuda.current_stream().synchronize()
            bad = [f for f in self._nan_flags if bool(f.any().item())]
            if wait or bad:
                self._nan_flags = []
            for f in bad:
                f.zero_()
            if bad:
                raise RuntimeError('NaN in loss')
            return
        if self._nan_event is None:
            return
        if wait:
            self._nan_event.synchronize()
        elif not self._nan_event.query():
            return
        bad = bool(self._nan_host.any().item())
        self._nan_event = None
        if bad:
            self._nan_dev.zero_()                # the kernels only ever SET the flags
            raise RuntimeError('NaN in loss')

    def _arm_nan(self, plan):
        if self.nan_check == 'off':
            return
        if plan.nan_host is not None:
            if not any(f is plan.nan_host for f in self._nan_flags):
                self._nan_flags.append(plan.nan_host)
            if self.nan_check == 'sync':
                self.check_nan(wait=True)
            return
        if self._nan_host is None or self._nan_host.numel() != plan.nan_dev.numel():
            self._nan_host = torch.zeros(plan.nan_dev.numel(), dtype=torch.int32).pin_memory()
        self._nan_host.copy_(plan.nan_dev, non_blocking=True)
        self._nan_dev = plan.nan_dev
        self._nan_event = torch.cuda.Event()
        self._nan_event.record()
        if self.nan_check == 'sync':
            self.check_nan(wait=True)

    def forward(self, x, target=None):
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        if target is None:
            plan = self._plan(x, False)
            plan.run_forward(x, None)
            if plan.eval_out is None:                  # no yolo layers (classifier cfgs): the last layer's output, reference :60-65
                f = plan.final
                if self.module_list[-1]._type == 'fc':
                    return f.raw.view(-1)[:f.C].clone()                                   # nn.Linear on x.view(input): (output,)
                return f.raw.view(f.B, f.H, f.W, f.ld)[..., :f.C].permute(0, 3, 1, 2).contiguous()
            return plan.eval_out.clone()
        self.check_nan(wait=False)
        target = tuple(t.to(x.device, torch.float32).contiguous() for t in target)
        plan = self._plan(x, True)
        anchor = self._flat.params[0]
        losses = _StepFn.apply(anchor, self, plan, x, target)      # loss, box, conf, cls over all heads; then each head's total
        self._arm_nan(plan)
        return losses


class DetectionModel(AnyModel):

    def forward(self, x, target=None):
        out = super().forward(x, target)
        if target is None:
            return out
        return {
            'loss': out[0],
            'giou_loss': out[1],
            'conf_loss': out[2],
            'class_loss': out[3],
            'loss_per_branch': list(out[4:]),
        }


class ClassifierModel(AnyModel):
    pass
