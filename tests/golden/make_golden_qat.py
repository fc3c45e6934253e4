#!/usr/bin/env python3
"""G9: QAT fixtures recorded by IMPORTING the reference (harness and rules: make_golden.py).

The reference's QAT model (BASELINE config 5) is ``DetectionModel(cfg, quant=True)`` -> ``tools.fuse_model`` ->
``tools.prepare_qat(backend='qnnpack')`` (``tools.py:256-290``, ``trainer.py:143-149``).  ``tools.fuse_model`` calls
``torch.quantization.fuse_modules``, which refuses train-mode modules on current PyTorch; as SURVEY 8(c) notes, the
train-mode twin ``fuse_modules_qat`` is substituted for that one call -- everything else runs unmodified.

Protocol: seeded fill of the FLOAT model (before fusing), then two forward/backward passes in train mode on a seeded
batch (no optimiser step: BatchNorm running statistics and the observers' moving averages change between the passes, the
weights do not), then one eval forward.  Recorded: losses of both passes, gradient norms of every parameter after pass 2
(+ three full gradient tensors), every observer's (min, max, scale, zero_point) after pass 2, running statistics of the
first / last BatchNorm, the eval output.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from seedfill import seeded_boxes, seeded_fill, seeded_images  # noqa: E402


def main():
    warnings.filterwarnings('ignore')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, interp, rloss, rparser = mg.import_reference()
    import torch.ao.quantization as tq
    torch.quantization.fuse_modules = lambda m, names, inplace=True: tq.fuse_modules_qat(m, names, inplace=inplace)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), '..'))
    from oracle import postprocess as opost

    npz = {}
    cfg = os.path.join(mg.REF, 'model/cfg/mobilenetv2-fpn.cfg')
    net = interp.DetectionModel(cfg, quant=True)
    seeded_fill(net, 0)
    net.train()
    tools.fuse_model(net)
    tools.prepare_qat(net, backend='qnnpack')

    B, S, ncls = 2, 256, 20      # 64x64 leaves 8 samples per channel in the stride-32 BatchNorms: with quantised inputs their
    # batch variance can be a few quantisation steps, and fp32 round-off differences between two CPUs change the loss by 10 %
    hw = np.array([[S // s, S // s] for s in (8, 16, 32)])
    losses = []
    for step in range(2):
        x = seeded_images(B, S, S, step)
        per = [opost.create_label(bb, hw, ncls) for bb in seeded_boxes(B, S, ncls, 1 + step)]
        tgt = tuple([torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
                    + [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)])
        net.zero_grad()
        out = net(x, tgt)
        out['loss'].mean().backward()
        losses.append([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                      + [v.item() for v in out['loss_per_branch']])
    npz['g9_loss'] = np.array(losses)
    names, gn = [], []
    for k, p in net.named_parameters():
        names.append(k)
        gn.append(p.grad.double().norm().item())
    npz['g9_gradnorm'] = np.array(gn)
    params = dict(net.named_parameters())
    npz['g9_grad_first_w'] = mg.t2n(params['module_list.0.conv.weight'].grad)
    npz['g9_grad_first_gamma'] = mg.t2n(params['module_list.0.conv.bn.weight'].grad)
    npz['g9_grad_mid_w'] = mg.t2n(params['module_list.40.conv.weight'].grad)
    npz['g9_grad_last_w'] = mg.t2n(params['module_list.101.conv.weight'].grad)
    npz['g9_grad_last_b'] = mg.t2n(params['module_list.101.conv.bias'].grad)
    # observers: (min, max, scale, zero_point) in module order
    obs_names, obs = [], []
    for k, m in net.named_modules():
        if type(m).__name__ == 'FusedMovingAvgObsFakeQuantize':
            obs_names.append(k)
            ap = m.activation_post_process
            obs.append([ap.min_val.item(), ap.max_val.item(), m.scale.item(), float(m.zero_point.item())])
    npz['g9_observers'] = np.array(obs, dtype=np.float64)
    sd = net.state_dict()
    rm = [k for k in sd if k.endswith('running_mean')]
    npz['g9_rmean_first'] = mg.t2n(sd[rm[0]])
    npz['g9_rvar_first'] = mg.t2n(sd[rm[0].replace('running_mean', 'running_var')])
    npz['g9_rmean_last'] = mg.t2n(sd[rm[-1]])
    npz['g9_rvar_last'] = mg.t2n(sd[rm[-1].replace('running_mean', 'running_var')])
    net.eval()
    with torch.no_grad():
        ev = net(seeded_images(B, S, S, 7))
    ev = ev[:, ::37]             # keep the fixture small: every 37th prediction row
    npz['g9_eval'] = mg.t2n(ev)
    np.savez_compressed(os.path.join(HERE, 'golden_qat.npz'), **npz)
    import json
    with open(os.path.join(HERE, 'g9_names.json'), 'w') as f:
        json.dump({'params': names, 'observers': obs_names,
                   'state_dict': [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]}, f)
    print('wrote', len(npz), 'arrays;', len(names), 'params,', len(obs_names), 'observers; losses', losses)


if __name__ == '__main__':
    main()
