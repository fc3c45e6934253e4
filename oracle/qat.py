"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the QAT path of BASELINE config 5.

The reference builds its QAT model out of PyTorch's own quantisation package (``tools.py:256-304``:
``fuse_modules`` -> ``prepare_qat(get_default_qat_qconfig('qnnpack'))``; ``model/parser.py:143-183`` FloatFunctional
add / cat; ``model/interpreter.py:27-28,42-43,52-53`` QuantStub / DeQuantStub).  What those library modules compute is
restated here with plain tensor ops:

* observer + fake-quantise (``FusedMovingAvgObsFakeQuantize`` -> ``torch.fused_moving_avg_obs_fake_quant``):
  moving-average min/max (constant 0.01, first batch initialises), qparams by ``ChooseQuantizationParams`` (range
  extended to contain 0, symmetric ranges centred for weights, scales below 6.1e-5 clipped), value
  ``(clamp(nearbyint(x/scale) + zp, qmin, qmax) - zp) * scale`` with a straight-through gradient inside the range.
  ``tests/test_oracle_qat.py`` checks this restatement against the library op itself on random tensors (the op is
  part of the installed torch, so that check also runs on the GPU box).
* fused conv + BN (+ReLU) QAT module (``torch.ao.nn.intrinsic.qat._ConvBnNd._forward_approximate``):
  ``sf = gamma / sqrt(running_var + eps)``; ``w_q = fq_w(w * sf)``; ``y = bn(conv(x, w_q) / sf)``; ReLU; ``fq_a``.
* heads: conv with ``fq_w(w)`` + bias, then ``fq_a``;  shortcut ``fq_a(a + b)`` then the activation;  concat route
  ``fq_a(cat)``;  input ``fq_a(x)``;  upsample and the single-source route are not observed.

Activations: quint8 [0, 255] per-tensor affine; weights: qint8 [-128, 127] per-tensor symmetric.
"""
import io
import math
from typing import Union

import torch
import torch.nn.functional as F
from torch import nn

from . import loss as oloss
from .net import OracleNet, read_specs

SMALL_SCALE = 6.1e-5
AVG_CONST = 0.01


def choose_qparams(lo: float, hi: float, qmin: int, qmax: int, symmetric: bool):
    """scale (float32 value) and integer zero point from a [lo, hi] range."""
    lo, hi = float(lo), float(hi)
    if lo < 0 and hi > 0 and symmetric:
        sym_min = -((qmax - qmin) // 2 + 1)
        sym_max = (qmax - qmin) // 2
        m = max(abs(lo / sym_min), abs(hi / sym_max))
        lo, hi = m * sym_min, m * sym_max
    lo, hi = min(lo, 0.0), max(hi, 0.0)
    # the library narrows to float at this point
    lo, hi = float(torch.tensor(lo, dtype=torch.float32)), float(torch.tensor(hi, dtype=torch.float32))
    scale = (hi - lo) / (qmax - qmin)
    fs = float(torch.tensor(scale, dtype=torch.float32))
    if fs == 0.0 or math.isinf(1.0 / fs):
        scale = 0.1
    if scale < SMALL_SCALE:
        org = scale
        scale = SMALL_SCALE
        if lo == 0.0:
            hi = SMALL_SCALE * (qmax - qmin)
        elif hi == 0.0:
            lo = -SMALL_SCALE * (qmax - qmin)
        else:
            amp = SMALL_SCALE / org
            lo, hi = lo * amp, hi * amp
    zp_min = qmin - lo / scale
    zp_max = qmax - hi / scale
    err_min = abs(qmin) - abs(lo / scale)
    err_max = abs(qmax) - abs(hi / scale)
    zp0 = zp_min if err_min < err_max else zp_max
    if lo < 0 and hi > 0 and symmetric:
        zp0 = (qmin + qmax) / 2
    if zp0 < qmin:
        zp = qmin
    elif zp0 > qmax:
        zp = qmax
    else:
        zp = int(torch.round(torch.tensor(zp0, dtype=torch.float64)).item())     # nearbyint: ties to even
    return float(torch.tensor(scale, dtype=torch.float32)), zp


class _FakeQuantSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, zp, qmin, qmax):
        inv = 1.0 / torch.tensor(scale, dtype=torch.float32)
        q = torch.round(x * inv.to(x.dtype)) + zp
        mask = (q >= qmin) & (q <= qmax)
        ctx.save_for_backward(mask)
        return (q.clamp(qmin, qmax) - zp) * scale

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None, None, None, None


class FakeQuant(nn.Module):
    """Observer + fake-quantiser with the library's buffers (so state_dicts line up): min_val, max_val, scale, zero_point."""

    def __init__(self, qmin: int, qmax: int, symmetric: bool):
        super().__init__()
        self.qmin, self.qmax, self.symmetric = qmin, qmax, symmetric
        self.register_buffer('min_val', torch.tensor(float('inf')))
        self.register_buffer('max_val', torch.tensor(float('-inf')))
        self.register_buffer('scale', torch.tensor([1.0]))
        self.register_buffer('zero_point', torch.tensor([0], dtype=torch.int32))
        self.observer_enabled = True
        self.fake_quant_enabled = True

    def forward(self, x):
        if self.observer_enabled:
            with torch.no_grad():
                lo, hi = torch.aminmax(x.detach().float())
                if math.isinf(self.min_val.item()):          # first batch initialises
                    self.min_val.copy_(lo)
                    self.max_val.copy_(hi)
                else:
                    self.min_val.add_(AVG_CONST * (lo - self.min_val))
                    self.max_val.add_(AVG_CONST * (hi - self.max_val))
                s, z = choose_qparams(self.min_val.item(), self.max_val.item(), self.qmin, self.qmax, self.symmetric)
                self.scale.fill_(s)
                self.zero_point.fill_(z)
        if not self.fake_quant_enabled:
            return x
        return _FakeQuantSTE.apply(x, float(self.scale.item()), int(self.zero_point.item()), self.qmin, self.qmax)


def act_fq():
    return FakeQuant(0, 255, False)


def weight_fq():
    return FakeQuant(-128, 127, True)


class OracleQATNet(nn.Module):
    """Same layer loop as :class:`oracle.net.OracleNet` with the QAT transformations applied (every activation is ReLU)."""

    def __init__(self, cfg: Union[str, io.IOBase]):
        super().__init__()
        self.float_net = OracleNet(cfg, quant_relu=True)
        n = len(self.float_net.module_list)
        self.input_fq = act_fq()
        self.w_fq = nn.ModuleList([weight_fq() if m.kind == 'convolutional' else nn.Identity() for m in self.float_net.module_list])
        self.a_fq = nn.ModuleList()
        for m in self.float_net.module_list:
            observed = m.kind == 'convolutional' or m.kind == 'shortcut' or \
                (m.kind == 'route' and isinstance(m.spec['layers'], list))
            self.a_fq.append(act_fq() if observed else nn.Identity())
        assert len(self.a_fq) == n

    @property
    def module_list(self):
        return self.float_net.module_list

    def forward(self, x, target=None):
        cache, heads = [], []
        x = self.input_fq(x)
        for i, m in enumerate(self.float_net.module_list):
            kind = m.kind
            if kind == 'convolutional':
                conv = m.conv
                if hasattr(m, 'bn'):
                    bn = m.bn
                    sf = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    wq = self.w_fq[i](conv.weight * sf.reshape(-1, 1, 1, 1))
                    y = F.conv2d(x, wq, None, conv.stride, conv.padding, conv.dilation, conv.groups)
                    y = bn(y / sf.reshape(1, -1, 1, 1))
                else:
                    y = F.conv2d(x, self.w_fq[i](conv.weight), conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
                if hasattr(m, 'act'):
                    y = m.act(y)
                x = self.a_fq[i](y)
            elif kind == 'shortcut':
                x = self.a_fq[i](x + cache[m.spec['from']])
                if m.act is not None:
                    x = m.act(x)
            elif kind == 'route':
                if isinstance(m.spec['layers'], list):
                    x = self.a_fq[i](torch.cat([cache[j] for j in m.idx], 1))
                else:
                    x = cache[m.idx[0]]
            elif kind == 'upsample':
                x = F.interpolate(x, scale_factor=m.spec['stride'], mode='nearest')
            elif kind == 'yolo':
                opt = m.opt
                x = OracleNet.decode(x, opt['classes'], opt['stride'])
                if target is not None:
                    pick = {8: (0, 3), 16: (1, 4), 32: (2, 5)}[opt['stride']]
                    x = oloss.loss_per_scale(x, target[pick[0]], target[pick[1]], opt)
                heads.append(x)
            else:
                raise ValueError('layer type %s is not part of the QAT detector cfgs' % kind)
            cache.append(x)
        if target is None:
            return torch.cat([h.reshape(h.shape[0], -1, h.shape[-1]) for h in heads], 1)
        total = [sum(parts) for parts in zip(*heads)]
        return {'loss': total[0], 'giou_loss': total[1], 'conf_loss': total[2], 'class_loss': total[3],
                'loss_per_branch': [sum(h[1:]) for h in heads]}
