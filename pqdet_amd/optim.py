"""Fused Adam for flat parameter storage -- drop-in for the reference's ``torch.optim.Adam`` use (``trainer.py:160-163``).

``Adam(model.parameters(), lr, weight_decay=...)`` keeps ``torch.optim.Optimizer`` semantics (``param_groups`` with a
writable ``'lr'`` -- the reference scheduler writes it every step, ``trainer.py:225`` -- ``zero_grad``, ``state_dict``), but
when every parameter of a group is a view into one flat buffer (which ``DetectionModel`` guarantees) the update is ONE
``pq_adam_step`` launch over the flat parameter / gradient / moment buffers (28 B per parameter of HBM traffic).
Same update rule as ``torch.optim.Adam`` (``amsgrad=False``): bias-corrected first/second moments, ``eps`` added after
``sqrt(v)/sqrt(bc2)``, L2 ``weight_decay`` folded into the gradient.
"""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    """Per-parameter state lives in ``self.state[p]`` under torch.optim.Adam's names (``step``, ``exp_avg``, ``exp_avg_sq``),
    so ``state_dict()`` / ``load_state_dict()`` save and restore the moments and are interchangeable with
    ``torch.optim.Adam``.  On the flat path the moment tensors are VIEWS into two flat buffers laid out like the parameter
    buffer; whenever they are not (first step, after ``load_state_dict``, after the model was re-flattened by ``.to()`` /
    ``.float()``) they are copied into place first, so the moments survive all of these."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale
        self._flat_bufs = {}          # group index -> (FlatParams, flat m, flat v)
        self._was_flat = set()
        self._linked = {}             # group index -> (id(FlatParams), state generation) of the last link
        self._state_gen = 0

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._state_gen += 1          # the loaded moment tensors are copies: re-link them on the next step

    def _group_flat(self, group):
        ps = group['params']
        owners = {getattr(p, '_pq_flat', (None, None))[0] for p in ps}
        if len(owners) != 1 or None in owners:
            return None
        flat = owners.pop()
        if not flat.still_valid() or len(ps) != len(flat.params):
            return None
        return flat

    def _link_flat_state(self, gi, flat):
        """Flat moment buffers for this group with every ``state[p]`` entry a view into them; returns (m, v, step)."""
        cur = self._flat_bufs.get(gi)
        if cur is None or cur[0] is not flat or cur[1].device != flat.flat.device:
            cur = (flat, torch.zeros_like(flat.flat), torch.zeros_like(flat.flat))
            self._flat_bufs[gi] = cur
        _, m, v = cur
        tag = (id(flat), self._state_gen)
        if self._linked.get(gi) == tag:                 # steady state: nothing moved since the last step
            return m, v, int(self.state[flat.params[0]]['step'])
        self._linked[gi] = tag
        step = 0
        for p, o in zip(flat.params, flat.offsets):
            st = self.state[p]
            mv, vv = m[o:o + p.numel()].view(p.shape), v[o:o + p.numel()].view(p.shape)
            old_m, old_v = st.get('exp_avg'), st.get('exp_avg_sq')
            if old_m is not None and old_m.data_ptr() != mv.data_ptr():       # loaded / carried over from another layout
                mv.copy_(old_m.to(mv.device, mv.dtype).view(p.shape))
                vv.copy_(old_v.to(vv.device, vv.dtype).view(p.shape))
            st['exp_avg'], st['exp_avg_sq'] = mv, vv
            step = max(step, int(st.get('step', 0)))
        return m, v, step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group['betas']
            flat = self._group_flat(group)
            if flat is not None and all(p.grad is not None and p.grad.data_ptr() == gv.data_ptr()
                                        for p, gv in zip(flat.params, flat.grad_views)):
                m, v, step = self._link_flat_state(gi, flat)
                step += 1
                for p in flat.params:
                    self.state[p]['step'] = step
                self._was_flat.add(gi)
                ops.adam_step(flat.flat, flat.grad, m, v, float(group['lr']), b1, b2, group['eps'],
                              group['weight_decay'], step, self.grad_scale)
                continue
            if gi in self._was_flat:
                import warnings
                warnings.warn('pqdet_amd.optim.Adam: parameter group %d left the fused flat-buffer path (parameters or gradients '
                              'were re-allocated); continuing per tensor with the same moments' % gi)
                self._was_flat.discard(gi)
                self._flat_bufs.pop(gi, None)
            for p in group['params']:           # generic path: one launch per tensor, same kernel
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not st['exp_avg'].is_contiguous() or st['exp_avg'].device != p.device:
                    st['exp_avg'] = st['exp_avg'].to(p.device).contiguous()
                    st['exp_avg_sq'] = st['exp_avg_sq'].to(p.device).contiguous()
                st['step'] = int(st['step']) + 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p.data, g, st['exp_avg'], st['exp_avg_sq'], float(group['lr']), b1, b2, group['eps'],
                              group['weight_decay'], st['step'], self.grad_scale)
        return loss


def sparse_l1(bn_modules, ratio: float):
    """Sparsity penalty of network-slimming training, applied between ``backward()`` and ``optimizer.step()``:
    ``gamma.grad += ratio * sign(gamma)`` for every collected BatchNorm (reference ``trainer.py:236-239``; the modules
    come from ``tools.get_bn_layers``).  One ``pq_sign_l1`` launch per BatchNorm on the gradient views."""
    for m in bn_modules:
        w = m.weight
        if w.grad is None:
            continue
        ops.sign_l1(w.grad, w.data, float(ratio))
