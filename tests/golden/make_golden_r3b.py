#!/usr/bin/env python3
"""Round-3 golden fixture G13, recorded by IMPORTING the reference (harness and rules: make_golden.py; only inputs, outputs
and structure are stored -- never reference text):

G13  the four classifier cfgs (``model/cfg/classifier/*.cfg``: 7 x 7 stem, maxpool, dense 3 x 3 convs, global avgpool, fc)
     through ``model.interpreter.AnyModel`` in eval mode: seeded fill (seed 5), one seeded 64 x 64 image (seed 7; ``FC``
     flattens with ``x.view(input)``, i.e. the reference itself only accepts one image) -> the (1000,) output, and for the
     ResNets the output of the maxpool layer (layer 1).

Writes ``golden_r3b.npz`` next to this file.
"""
import glob
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from seedfill import seeded_fill, seeded_images  # noqa: E402


def main():
    warnings.filterwarnings('ignore')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, interp, rloss, rparser = mg.import_reference()
    npz = {}
    x = seeded_images(1, 64, 64, 7)
    for path in sorted(glob.glob(os.path.join(mg.REF, 'model/cfg/classifier/*.cfg'))):
        name = os.path.splitext(os.path.basename(path))[0]
        net = interp.AnyModel(path)
        seeded_fill(net, 5)
        net.eval()
        cache = []
        hooks = [m.register_forward_hook(lambda mod, inp, out, c=cache: c.append(out)) for m in net.module_list]
        with torch.no_grad():
            y = net(x)
        for h in hooks:
            h.remove()
        assert y.shape == (1000,), y.shape
        npz['g13_%s_out' % name] = mg.t2n(y)
        if net.module_list[1]._type == 'maxpool':
            npz['g13_%s_maxpool' % name] = mg.t2n(cache[1])
        print(name, 'out', float(y.abs().mean()), 'layers', len(net.module_list))
    np.savez_compressed(os.path.join(HERE, 'golden_r3b.npz'), **npz)


if __name__ == '__main__':
    main()
