"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PyTorch-CPU fp32 restatement of the reference network path:

* layer construction          -- reference ``model/parser.py:361-470`` (``Parser.torch_layers``)
* layer loop / target routing -- reference ``model/interpreter.py:38-85`` (``AnyModel.forward``,
  ``DetectionModel.forward``, ``_TARGET_MAP`` :16-20)
* head decode                 -- reference ``model/parser.py:185-235``
* loss                        -- ``oracle/loss.py`` (reference ``model/loss.py:7-115``)

The cfg text is read with the product's host-side reader (``pqdet_amd.model.cfgtext``; string processing only,
itself pinned by golden fixture G1); every tensor operation below is plain ``torch`` on the CPU.  The module tree
uses the reference's child names so that ``state_dict`` keys line up with the product model
(``module_list.<i>.conv.weight`` ...) and weights can be exchanged with ``load_state_dict``.
"""
import io
from typing import List, Union

import torch
import torch.nn.functional as F
from torch import nn

from pqdet_amd.model import cfgtext

from . import loss as oloss

_ACTS = {
    'logistic': nn.Sigmoid,
    'leaky': lambda: nn.LeakyReLU(0.1),
    'relu': nn.ReLU,
    'relu6': nn.ReLU6,
    'tanh': nn.Tanh,
}


class _Holder(nn.Module):
    """parameter-less layer (shortcut / route / upsample / yolo / pools)."""

    def __init__(self, kind, spec):
        super().__init__()
        self.kind = kind
        self.spec = spec


def read_specs(cfg: Union[str, io.IOBase]) -> List[dict]:
    if isinstance(cfg, str):
        with open(cfg, 'r') as f:
            return list(cfgtext.iter_sections(f))
    return list(cfgtext.iter_sections(cfg))


class OracleNet(nn.Module):
    def __init__(self, cfg, quant_relu: bool = False):
        super().__init__()
        specs = read_specs(cfg)
        chans, stride = None, 1
        out_ch, strides, layers = [], [], []
        for s in specs:
            kind = s['name']
            if kind == 'net':
                chans = s['channels']
                continue
            if kind == 'convolutional':                       # model/parser.py:389-410
                k = s['size']
                pad = k // 2 if bool(s['pad']) else s['padding']
                use_bn = s['batch_normalize'] != 0
                m = nn.Sequential()
                m.add_module('conv', nn.Conv2d(chans, s['filters'], k, s['stride'], pad,
                                               groups=s['groups'], bias=not use_bn))
                if use_bn:
                    m.add_module('bn', nn.BatchNorm2d(s['filters']))
                if s['activation'] != 'linear':
                    m.add_module('act', nn.ReLU() if quant_relu else _ACTS[s['activation']]())
                m.kind = kind
                chans = s['filters']
                if stride is not None:                        # LayerInfo.update_stride: no-op after an avgpool (:26-29)
                    stride *= s['stride']
            elif kind == 'shortcut':                          # :415-424
                assert out_ch[-1] == out_ch[s['from']]
                m = _Holder(kind, s)
                m.act = _ACTS[s['activation']]() if s['activation'] != 'linear' else None
            elif kind == 'scale_channels':                    # :425-429
                m = _Holder(kind, s)
                stride = strides[s['from']]
            elif kind == 'route':                             # :430-440
                idx = s['layers'] if isinstance(s['layers'], list) else [s['layers']]
                m = _Holder(kind, s)
                m.idx = idx
                chans = sum(out_ch[i] for i in idx)
                stride = strides[idx[0]]
            elif kind == 'upsample':                          # :450-452
                m = _Holder(kind, s)
                if stride is not None:
                    stride //= s['stride']
            elif kind == 'fc':                                # :122-137 FC: nn.Linear on x.view(input), then the activation
                m = _Holder(kind, s)
                m.fc = nn.Linear(s['input'], s['output'])
                m.act = _ACTS[s['activation']]() if s['activation'] != 'linear' else None
                chans = s['output']
            elif kind == 'maxpool':
                m = _Holder(kind, s)
                if stride is not None:
                    stride *= s['stride']
            elif kind == 'avgpool':
                m = _Holder(kind, s)
                stride = None
            elif kind == 'yolo':                              # :453-459
                m = _Holder(kind, s)
                m.opt = dict(s, stride=stride)
            else:
                raise ValueError(kind)
            out_ch.append(chans)
            strides.append(stride)
            layers.append(m)
        self.module_list = nn.ModuleList(layers)
        self.strides = strides

    # ---- head decode, model/parser.py:206-235 ----
    @staticmethod
    def decode(raw, num_classes, stride):
        b, _, h, w = raw.shape
        r = raw.permute(0, 2, 3, 1).reshape(b, h, w, -1, 5 + num_classes)
        cx = (torch.arange(w, dtype=torch.float32) + 0.5).view(1, 1, w, 1)
        cy = (torch.arange(h, dtype=torch.float32) + 0.5).view(1, h, 1, 1)
        centre = torch.stack([cx.expand(1, h, w, 1), cy.expand(1, h, w, 1)], -1)   # (1,H,W,1,2) = (x, y)
        lo = (centre - torch.exp(r[..., 0:2])) * stride
        hi = (centre + torch.exp(r[..., 2:4])) * stride
        return torch.cat([lo, hi, torch.sigmoid(r[..., 4:5]), torch.sigmoid(r[..., 5:])], -1)

    # ---- layer loop, model/interpreter.py:38-85 ----
    def forward(self, x, target=None):
        cache, heads = [], []
        for m in self.module_list:
            kind = m.kind
            if kind == 'convolutional':
                x = m(x)
            elif kind == 'shortcut':
                x = x + cache[m.spec['from']]     # reference adds in place; same values
                if m.act is not None:
                    x = m.act(x)
            elif kind == 'scale_channels':
                x = cache[m.spec['from']] * x
            elif kind == 'route':
                x = cache[m.idx[0]] if len(m.idx) == 1 and not isinstance(m.spec['layers'], list) \
                    else torch.cat([cache[i] for i in m.idx], 1)
            elif kind == 'upsample':
                x = F.interpolate(x, scale_factor=m.spec['stride'], mode='nearest')
            elif kind == 'maxpool':
                k = m.spec['size']
                pad = k // 2 if bool(m.spec['pad']) else m.spec['padding']
                x = F.max_pool2d(x, k, m.spec['stride'], pad)
            elif kind == 'avgpool':
                x = F.adaptive_avg_pool2d(x, (m.spec['height'], m.spec['width']))
            elif kind == 'fc':
                x = m.fc(x.view(m.spec['input']))
                if m.act is not None:
                    x = m.act(x)
            elif kind == 'yolo':
                opt = m.opt
                x = self.decode(x, opt['classes'], opt['stride'])
                if target is not None:
                    pick = {8: (0, 3), 16: (1, 4), 32: (2, 5)}[opt['stride']]   # interpreter.py:16-20
                    x = oloss.loss_per_scale(x, target[pick[0]], target[pick[1]], opt)
                heads.append(x)
            else:
                raise ValueError(kind)
            cache.append(x)
        if not heads:                                         # model/interpreter.py:60-65: no output layer -> the last layer's output
            return cache[-1]
        if target is None:
            return torch.cat([h.reshape(h.shape[0], -1, h.shape[-1]) for h in heads], 1)
        total = [sum(parts) for parts in zip(*heads)]
        return {
            'loss': total[0], 'giou_loss': total[1], 'conf_loss': total[2], 'class_loss': total[3],
            'loss_per_branch': [sum(h[1:]) for h in heads],
        }
