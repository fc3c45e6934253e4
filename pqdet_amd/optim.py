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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale
        self._flat_state = {}

    def _group_flat(self, group):
        ps = group['params']
        owners = {getattr(p, '_pq_flat', (None, None))[0] for p in ps}
        if len(owners) != 1 or None in owners:
            return None
        flat = owners.pop()
        if not flat.still_valid() or len(ps) != len(flat.params):
            return None
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group['betas']
            flat = self._group_flat(group)
            if flat is not None and all(p.grad is not None and p.grad.data_ptr() == gv.data_ptr()
                                        for p, gv in zip(flat.params, flat.grad_views)):
                st = self._flat_state.get(gi)
                if st is None or st['m'].numel() != flat.flat.numel() or st['m'].device != flat.flat.device:
                    st = {'step': 0, 'm': torch.zeros_like(flat.flat), 'v': torch.zeros_like(flat.flat)}
                    self._flat_state[gi] = st
                st['step'] += 1
                ops.adam_step(flat.flat, flat.grad, st['m'], st['v'], float(group['lr']), b1, b2, group['eps'],
                              group['weight_decay'], st['step'], self.grad_scale)
                continue
            for p in group['params']:           # generic path: one launch per tensor, same kernel
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['m'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['v'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p.data, g, st['m'], st['v'], float(group['lr']), b1, b2, group['eps'],
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
