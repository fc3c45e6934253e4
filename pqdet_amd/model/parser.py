"""cfg -> module list.  Host-side mirror of the reference ``model/parser.py``.

Same public names and argument meaning as the reference (``Parser(fp).torch_layers(quant, onnx)``,
``YOLOLayer(opt, onnx)``, ``Decode``, ``build_center_grid``, ``ShortCut``, ``Route``, ``ScaleChannels``,
``FC``, ``ACTIVATION_MAP``, ``DEFAULT_LAYERS``, ``str2value``), same module attributes
(``_type/_stride/_output_channels/_raw/_from/_layers/_notprune``), same child names (``conv/bn/act``)
and therefore the same ``state_dict`` key layout (``module_list.<i>.conv.weight`` ...), see
reference ``model/parser.py:361-470``.

What differs is *who computes*: the modules built here are parameter/attribute holders.  The arithmetic of a
training or inference step is executed by :mod:`pqdet_amd.engine` (hand-written gfx950 kernels behind the
C-ABI in ``include/pqdet_hip.h``), driven by ``DetectionModel.forward``.  ``YOLOLayer``/``Decode`` keep a
working standalone ``forward`` (HIP decode + HIP loss) because the reference's NAS code calls them directly
(``nas/detnet.py:106-108``).
"""
from typing import IO, List, Union

import torch
from torch import nn

from . import cfgtext
from .cfgtext import str2value  # noqa: F401  (re-export, reference name)

DEFAULT_LAYERS = cfgtext.SECTION_DEFAULTS

# activation name -> (module factory, engine activation code).  reference model/parser.py:106-112
ACT_LINEAR, ACT_RELU, ACT_RELU6, ACT_LEAKY, ACT_LOGISTIC, ACT_TANH = range(6)
ACTIVATION_MAP = {
    'logistic': nn.Sigmoid,
    'leaky': lambda: nn.LeakyReLU(0.1, inplace=True),
    'relu': lambda: nn.ReLU(inplace=True),
    'relu6': lambda: nn.ReLU6(inplace=True),
    'tanh': nn.Tanh,
}
ACT_CODE = {'linear': ACT_LINEAR, 'relu': ACT_RELU, 'relu6': ACT_RELU6, 'leaky': ACT_LEAKY,
            'logistic': ACT_LOGISTIC, 'tanh': ACT_TANH}


def act_code_of(module) -> int:
    """Engine activation code of an ``act`` child (handles the quant ReLU override too)."""
    if module is None:
        return ACT_LINEAR
    if isinstance(module, nn.ReLU6):
        return ACT_RELU6
    if isinstance(module, nn.ReLU):
        return ACT_RELU
    if isinstance(module, nn.LeakyReLU):
        return ACT_LEAKY
    if isinstance(module, nn.Sigmoid):
        return ACT_LOGISTIC
    if isinstance(module, nn.Tanh):
        return ACT_TANH
    raise ValueError('unsupported activation module %r' % (module,))


class LayerInfo:
    """Running (channels, size, stride) bookkeeping while walking the cfg; reference :17-41."""

    def __init__(self):
        self.shape = [None, None, None]
        self.stride = 1

    def update_stride(self, stride, down=False):
        if self.stride is None:
            return
        self.stride = self.stride // stride if down else self.stride * stride

    def set_size(self, size):
        self.shape[1:] = size

    @property
    def channels(self):
        return self.shape[0]

    @channels.setter
    def channels(self, c):
        self.shape[0] = c


class ConvBlock(nn.Sequential):
    """``Sequential(conv[, bn][, act])`` parameter holder (reference :389-410).

    The arithmetic runs in the engine; calling the block on its own is not a supported path (it would
    silently fall back to a vendor library), so it fails loudly instead."""

    def forward(self, x):  # pragma: no cover - guard
        raise RuntimeError('pqdet_amd conv blocks are executed by DetectionModel/engine, not standalone')


class FC(nn.Module):
    """reference :122-137 (classifier cfgs only)."""

    def __init__(self, input: int, output: int, activation: str):
        super().__init__()
        self.fc = nn.Linear(input, output)
        self.act = ACTIVATION_MAP[activation]() if activation != 'linear' else None
        self.input = input

    def forward(self, x):
        x = self.fc(x.view(self.input))
        return x if self.act is None else self.act(x)


class ShortCut(nn.Module):
    """reference :139-156.  In-place ``x += other`` then optional activation."""

    def __init__(self, activation: str, quant: bool = False):
        super().__init__()
        self.quant = quant
        self.act = ACTIVATION_MAP[activation]() if activation != 'linear' else None

    def forward(self, x, other):
        x += other
        return x if self.act is None else self.act(x)


class ScaleChannels(nn.Module):
    """reference :158-168."""

    def __init__(self, quant: bool = False):
        super().__init__()
        self.quant = quant

    def forward(self, x, other):
        return other * x


class Route(nn.Module):
    """reference :170-183: alias (one index) or channel concat."""

    def __init__(self, quant: bool = False, single: bool = False):
        super().__init__()
        self.quant = quant
        self.single = single

    def forward(self, xs):
        return xs[0] if self.single else torch.cat(xs, dim=1)


def build_center_grid(height, width) -> torch.Tensor:
    """(H, W, 1, 2) cell centres, ``[...,0] = col + .5``, ``[...,1] = row + .5`` (reference :185-192)."""
    ys = torch.arange(0, height, dtype=torch.float32) + 0.5
    xs = torch.arange(0, width, dtype=torch.float32) + 0.5
    gx = xs.view(1, width).expand(height, width)
    gy = ys.view(height, 1).expand(height, width)
    return torch.stack([gx, gy], dim=-1).unsqueeze(2).contiguous()


class Decode(nn.Module):
    """Head decode (reference :194-235): NCHW raw head -> (B,H,W,A,5+C) boxes/conf/prob.

    Runs the ``pq_decode`` HIP kernel; differentiable through :class:`pqdet_amd.ops.DecodeFn`."""

    def __init__(self, num_classes: int, stride: int, onnx: bool = False):
        super().__init__()
        self.num_classes = num_classes
        self.stride = stride
        self.onnx = onnx
        self.grid_size = (0, 0)   # kept for attribute compatibility; the kernel computes centres inline

    def forward(self, conv):
        from .. import ops
        return ops.decode_nchw(conv, self.num_classes, self.stride)


class YOLOLayer(nn.Module):
    """reference :237-249."""

    def __init__(self, opt: dict, onnx: bool = False):
        super().__init__()
        self.decode = Decode(opt['classes'], opt['stride'], onnx)
        self.opt = opt

    def forward(self, x, target=None):
        x = self.decode(x)
        if target is None:
            return x
        from .loss import loss_per_scale
        return loss_per_scale(x, *target, self.opt)


def _solve_padding(size: int, padding: int, pad) -> int:
    return size // 2 if bool(pad) else padding


class Parser:
    """``Parser(fp)``: ``fp`` is any iterable of text lines (open file, ``StringIO``)."""

    def __init__(self, fp: IO):
        self.fp = fp

    def gen_layers(self):
        return cfgtext.iter_sections(self.fp)

    def torch_layers(self, quant: bool = False, onnx: bool = False) -> List[nn.Module]:
        info = LayerInfo()
        mods: List[nn.Module] = []

        def pos(where: int = 0):          # 1-based position used in assertion texts (reference :363-364)
            return len(mods) + where + 1 if where <= 0 else where + 1

        def need_same_channels(name, a, b):
            ca, cb = mods[a]._output_channels, mods[b]._output_channels
            assert ca == cb, '{} layer[{}]: out channels dont match between layer {}({}) and {}({})'.format(
                name, pos(), pos(a), ca, pos(b), cb)

        def need_same_strides(name, idxs):
            s0 = mods[idxs[0]]._stride
            assert all(mods[i]._stride == s0 for i in idxs), \
                '{} layer[{}]: not all input strides is same: {}'.format(
                    name, pos(), [mods[i]._stride for i in idxs])

        for spec in self.gen_layers():
            name = spec['name']
            if name == 'net':
                info.channels = spec['channels']
                continue
            if name == 'convolutional':
                m = ConvBlock()
                has_bias = spec['batch_normalize'] == 0
                m.add_module('conv', nn.Conv2d(
                    info.channels, spec['filters'], kernel_size=spec['size'], stride=spec['stride'],
                    padding=_solve_padding(spec['size'], spec['padding'], spec['pad']),
                    groups=spec['groups'], bias=has_bias))
                info.channels = spec['filters']
                info.update_stride(spec['stride'])
                if not has_bias:
                    m.add_module('bn', nn.BatchNorm2d(spec['filters']))
                if spec['activation'] != 'linear':
                    act = ACTIVATION_MAP[spec['activation']]()
                    m.add_module('act', nn.ReLU() if quant else act)
            elif name == 'fc':
                m = FC(spec['input'], spec['output'], spec['activation'])
                info.channels = spec['output']
                mods[-1]._notprune = True
            elif name == 'shortcut':
                need_same_channels(name, -1, spec['from'])
                m = ShortCut(spec['activation'], quant)
                m._from = spec['from']
                mods[-1]._notprune = True
                mods[spec['from']]._notprune = True
            elif name == 'scale_channels':
                need_same_channels(name, -1, spec['from'])
                m = ScaleChannels(quant)
                m._from = spec['from']
                info.stride = mods[spec['from']]._stride
            elif name == 'route':
                idxs = spec['layers']
                single = isinstance(idxs, int)
                if single:
                    idxs = [idxs]
                need_same_strides(name, idxs)
                m = Route(quant, single)
                m._layers = idxs
                info.channels = sum(mods[i]._output_channels for i in idxs)
                info.stride = mods[idxs[0]]._stride
            elif name == 'maxpool':
                m = nn.MaxPool2d(spec['size'], stride=spec['stride'],
                                 padding=_solve_padding(spec['size'], spec['padding'], spec['pad']))
                info.update_stride(spec['stride'])
            elif name == 'avgpool':
                size = (spec['height'], spec['width'])
                m = nn.AdaptiveAvgPool2d(size)
                info.stride = None
                info.set_size(size)
            elif name == 'upsample':
                m = nn.UpsamplingNearest2d(scale_factor=spec['stride'])
                info.update_stride(spec['stride'], down=True)
            elif name == 'yolo':
                assert spec['bbox_loss'] in {'diou', 'ciou', 'giou', 'iou', 'l1'}, \
                    'unspport bbox loss type in yolo layer: {}'.format(spec['bbox_loss'])
                opt = spec.copy()
                opt['stride'] = info.stride
                m = YOLOLayer(opt, onnx)
                mods[-1]._notprune = True
            elif name == 'dropout':
                m = nn.Dropout(p=spec['probability'])
            else:
                raise ValueError(f"unsupport layer type: '{name}'")
            m._output_channels = info.channels
            m._stride = info.stride
            m._type = name
            m._raw = spec
            mods.append(m)
        return mods
