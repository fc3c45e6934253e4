#!/usr/bin/env python3
"""Round-3 golden fixture recorded by IMPORTING the reference (harness and rules: make_golden.py; only inputs, outputs and
structure are stored -- never reference text):

G5b-pan  WELL-CONDITIONED end-to-end gradients of regnetx-600m-pan (BASELINE config 4's network) at 512x512 batch 2,
         ``model.eval()`` + targets (BatchNorm on its running statistics, ``model/interpreter.py:72-85``), same protocol as
         G5b of make_golden_r2.py: losses, the gradient norm of every parameter and the FULL gradient tensors of parameters
         chosen by ROLE from the layer list: stem, first backbone block, a projection shortcut (strided 1x1), both stride-2
         grouped convs of the bottom-up path, the conv behind every two-input route (both fan-ins of the top-down and of the
         bottom-up path), the three head convs.

Writes ``golden_r3.npz`` and ``g_r3_names.json`` next to this file.
"""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from seedfill import seeded_boxes, seeded_fill, seeded_images  # noqa: E402

MAX_ELEMS = 150_000


def pick_parameters(net):
    """Parameter names by role (see the module docstring); weights above MAX_ELEMS elements are represented by their
    BatchNorm bias (whose gradient is the column sum of the same dy')."""
    mods = list(net.module_list)
    picks = []

    def add(i, what='weight'):
        m = mods[i]
        if what == 'weight' and m.conv.weight.numel() <= MAX_ELEMS:
            picks.append('module_list.%d.conv.weight' % i)
        elif hasattr(m, 'bn'):
            picks.append('module_list.%d.bn.bias' % i)
        else:
            picks.append('module_list.%d.conv.bias' % i)

    convs = [i for i, m in enumerate(mods) if m._type == 'convolutional']
    add(convs[0]); picks.append('module_list.%d.bn.weight' % convs[0])
    add(convs[1])
    first_g = next(i for i in convs if mods[i].conv.groups > 1)
    add(first_g)
    proj = next(i for i in convs[1:] if mods[i].conv.kernel_size[0] == 1 and mods[i].conv.stride[0] == 2)
    add(proj)
    first_head = next(i for i, m in enumerate(mods) if m._type == 'yolo')
    for i in convs:
        if i > first_head and mods[i].conv.groups > 1 and mods[i].conv.stride[0] == 2:
            add(i)                                              # stride-2 grouped convs of the bottom-up path
    for i, m in enumerate(mods):
        if m._type == 'route' and not isinstance(m._layers, int) and len(m._layers) == 2 and mods[i + 1]._type == 'convolutional':
            add(i + 1)                                          # conv behind a two-input concatenation
        if m._type == 'yolo':
            add(i - 1)
            picks.append('module_list.%d.conv.bias' % (i - 1))
    out = []
    for k in picks:
        if k not in out:
            out.append(k)
    return out


def main():
    warnings.filterwarnings('ignore')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, interp, rloss, rparser = mg.import_reference()
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    from oracle import postprocess as opost

    def targets(batch, size, ncls, seed):
        hw = np.array([[size // s, size // s] for s in (8, 16, 32)])
        per = [opost.create_label(bb, hw, ncls) for bb in seeded_boxes(batch, size, ncls, seed)]
        return tuple([torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
                     + [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)])

    npz = {}
    net = interp.DetectionModel(os.path.join(mg.REF, 'model/cfg', 'regnetx-600m-pan.cfg'))
    seeded_fill(net, 5)
    net.eval()
    out = net(seeded_images(2, 512, 512, 5), targets(2, 512, 20, 6))
    out['loss'].mean().backward()
    npz['g5b_rxpan_loss'] = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                                     + [v.item() for v in out['loss_per_branch']])
    params = dict(net.named_parameters())
    npz['g5b_rxpan_gradnorm'] = np.array([p.grad.double().norm().item() for p in params.values()])
    full = pick_parameters(net)
    for k in full:
        npz['g5b_rxpan_grad_%s' % k] = mg.t2n(params[k].grad)
    print('G5b-pan losses', npz['g5b_rxpan_loss'])
    print('full gradient tensors:', [(k, tuple(params[k].shape)) for k in full])
    np.savez_compressed(os.path.join(HERE, 'golden_r3.npz'), **npz)
    with open(os.path.join(HERE, 'g_r3_names.json'), 'w') as f:
        json.dump({'g5b': {'rxpan': list(params)}, 'g5b_full': {'rxpan': full}}, f)
    print('wrote', len(npz), 'arrays')


if __name__ == '__main__':
    main()
