"""Programmatic cfg text for the BASELINE networks.

The reference ships its topologies as ``model/cfg/*.cfg`` text files; those files do not travel with this repo.
The generators below emit equivalent darknet-style cfg text from the published architecture tables
(MobileNetV2: Sandler et al. 2018, table 2; RegNetX-600MF: Radosavovic et al. 2020) plus the FPN / PAN YOLO
necks, and ``tests/test_parser.py`` checks that the emitted text builds exactly the layer list recorded from the
reference's own files (fixture G1: types, channels, strides, groups, activations, route/shortcut indices, yolo
options, parameter count).  Any user-supplied ``.cfg`` goes through the same :class:`~pqdet_amd.model.parser.Parser`.
"""
from typing import Callable, Dict, List


class _Cfg:
    def __init__(self, channels=3):
        self.lines: List[str] = ['[net]', 'channels=%d' % channels, '']
        self.n = 0          # number of layers emitted so far (index of the next layer)

    def _emit(self, section: str, **kv) -> int:
        self.lines.append('[%s]' % section)
        for k, v in kv.items():
            if isinstance(v, (list, tuple)):
                v = ', '.join(str(x) for x in v)
            self.lines.append('%s=%s' % (k, v))
        self.lines.append('')
        self.n += 1
        return self.n - 1

    def conv(self, filters, size=1, stride=1, groups=1, act='relu6', bn=True) -> int:
        kv = dict(filters=filters, size=size, stride=stride, pad=1)
        if groups != 1:
            kv['groups'] = groups
        if bn:
            kv['batch_normalize'] = 1
        kv['activation'] = act
        return self._emit('convolutional', **kv)

    def shortcut(self, frm, act='linear') -> int:
        return self._emit('shortcut', **{'from': frm, 'activation': act})

    def route(self, layers) -> int:
        return self._emit('route', layers=layers)

    def upsample(self, stride=2) -> int:
        return self._emit('upsample', stride=stride)

    def yolo(self, classes, bbox_loss='l1', l1_loss_gain=0.05, ignore_thresh=.5) -> int:
        return self._emit('yolo', classes=classes, ignore_thresh=ignore_thresh, bbox_loss=bbox_loss,
                          l1_loss_gain=l1_loss_gain)

    def text(self) -> str:
        return '\n'.join(self.lines) + '\n'


def _sep_block(c: _Cfg, width: int, act: str) -> None:
    """1x1 width -> depthwise 3x3 width -> 1x1 2*width (the neck's repeated unit)."""
    c.conv(width, 1, act=act)
    c.conv(width, 3, groups=width, act=act)
    c.conv(2 * width, 1, act=act)


def _fpn_head(c: _Cfg, taps: List[int], widths: List[int], classes: int, act: str) -> None:
    """Top-down FPN with three YOLO heads.  ``taps`` = backbone layer indices at stride 16 and 8."""
    per_anchor = 3 * (5 + classes)
    for level, width in enumerate(widths):
        for _ in range(3):
            _sep_block(c, width, act)
        c.conv(per_anchor, 1, act='linear', bn=False)
        c.yolo(classes)
        if level == len(widths) - 1:
            break
        c.route(-5)
        c.conv(widths[level + 1], 1, act=act)
        c.upsample(2)
        c.route([-1, taps[level]])


def mobilenetv2_fpn(classes: int = 20) -> str:
    c = _Cfg()
    act = 'relu6'
    c.conv(32, 3, stride=2, act=act)
    c.conv(32, 3, groups=32, act=act)
    c.conv(16, 1, act='linear')
    cin = 16
    stage_end = {}
    stride_now = 2
    # (expansion, out channels, repeats, first stride)
    for t, cout, n, s in ((6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)):
        for i in range(n):
            st = s if i == 0 else 1
            c.conv(cin * t, 1, act=act)
            c.conv(cin * t, 3, stride=st, groups=cin * t, act=act)
            last = c.conv(cout, 1, act='linear')
            if st == 1 and cin == cout:
                last = c.shortcut(-4)
            cin = cout
            stride_now *= st
        stage_end[(cout, stride_now)] = last
    c.conv(1280, 1, act=act)
    taps = [stage_end[(96, 16)], stage_end[(32, 8)]]
    _fpn_head(c, taps, [512, 256, 128], classes, act)
    return c.text()


def _xblock(c: _Cfg, width: int, stride: int, group_width: int, project: bool) -> int:
    """RegNet X block: 1x1 -> grouped 3x3 (stride) -> 1x1 (linear) + identity / strided 1x1 projection, then ReLU."""
    if project:
        c.conv(width, 1, stride=stride, act='linear')      # projection of the block input, consumed by the shortcut
        c.route(-2)                                        # back to the block input
    c.conv(width, 1, act='relu')
    c.conv(width, 3, stride=stride, groups=width // group_width, act='relu')
    c.conv(width, 1, act='linear')
    return c.shortcut(-5 if project else -4, act='relu')


def _regnetx_600m_backbone(c: _Cfg) -> Dict[int, int]:
    """RegNetX-600MF (Radosavovic et al. 2020): stem 32, depths 1-3-5-7, widths 48-96-240-528, group width 24.
    Returns {stride: index of the last layer at that stride}."""
    c.conv(32, 3, stride=2, act='relu')
    cin, stride_now, ends = 32, 2, {}
    for depth, width in ((1, 48), (3, 96), (5, 240), (7, 528)):
        for i in range(depth):
            st = 2 if i == 0 else 1
            last = _xblock(c, width, st, 24, project=(st != 1 or cin != width))
            cin = width
            stride_now *= st
        ends[stride_now] = last
    return ends


def regnetx_600m_fpn(classes: int = 20) -> str:
    c = _Cfg()
    ends = _regnetx_600m_backbone(c)
    per_anchor = 3 * (5 + classes)
    cin = 528
    widths, taps = (352, 176, 80), (ends[16], ends[8])
    for level, width in enumerate(widths):
        for i in range(3):
            _xblock(c, width, 1, 16, project=(i == 0))
        c.conv(per_anchor, 1, act='linear', bn=False)
        c.yolo(classes)
        if level == len(widths) - 1:
            break
        c.route(-7)
        c.conv(widths[level + 1], 1, act='relu')
        c.upsample(2)
        c.route([-1, taps[level]])
    return c.text()


def _gsep(c: _Cfg, width: int, stride: int = 1) -> int:
    """1x1 -> grouped 3x3 (group width 16) -> 1x1, all ReLU (the PAN neck's unit)."""
    c.conv(width, 1, act='relu')
    c.conv(width, 3, stride=stride, groups=width // 16, act='relu')
    return c.conv(width, 1, act='relu')


def regnetx_600m_pan(classes: int = 20) -> str:
    c = _Cfg()
    ends = _regnetx_600m_backbone(c)
    per_anchor = 3 * (5 + classes)
    # top-down
    p5 = _gsep(c, 352)
    c.upsample(2)
    c.route([-1, ends[16]])
    p4 = _gsep(c, 176)
    c.upsample(2)
    c.route([-1, ends[8]])
    _gsep(c, 96)
    c.conv(per_anchor, 1, act='linear', bn=False)
    c.yolo(classes)
    # bottom-up
    for width, lateral in ((176, p4), (352, p5)):
        c.route(-3)
        _gsep(c, width, stride=2)
        c.route([-1, lateral - c.n])
        _gsep(c, width)
        c.conv(per_anchor, 1, act='linear', bn=False)
        c.yolo(classes)
    return c.text()


_ZOO: Dict[str, Callable[..., str]] = {
    'mobilenetv2-fpn': mobilenetv2_fpn,
    'regnetx-600m-fpn': regnetx_600m_fpn,
    'regnetx-600m-pan': regnetx_600m_pan,
}


def available() -> List[str]:
    return sorted(_ZOO)


def cfg_text(name: str, **kw) -> str:
    """cfg text for a zoo network, e.g. ``DetectionModel(io.StringIO(cfg_text('mobilenetv2-fpn')))``."""
    return _ZOO[name](**kw)
