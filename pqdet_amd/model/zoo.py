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


_ZOO: Dict[str, Callable[..., str]] = {
    'mobilenetv2-fpn': mobilenetv2_fpn,
}


def available() -> List[str]:
    return sorted(_ZOO)


def cfg_text(name: str, **kw) -> str:
    """cfg text for a zoo network, e.g. ``DetectionModel(io.StringIO(cfg_text('mobilenetv2-fpn')))``."""
    return _ZOO[name](**kw)
