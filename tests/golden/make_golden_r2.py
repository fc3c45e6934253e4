#!/usr/bin/env python3
"""Round-2 golden fixtures recorded by IMPORTING the reference (harness and rules: make_golden.py; only inputs, outputs and
structure are stored -- never reference text):

G4b  ``tools.ciou`` on NaN-free box pairs (every box has positive width / height and no pair is identical)
G5b  WELL-CONDITIONED end-to-end gradients: mobilenetv2-fpn and regnetx-600m-fpn, 512x512 batch 2 (BASELINE config-1
     shape), ``model.eval()`` + targets (BatchNorm normalises with its running statistics, ``model/interpreter.py:72-85``):
     losses, the gradient norm of every parameter and the FULL gradient tensors of parameters spread over stem / backbone /
     neck / heads.  Without batch statistics fp32 round-off is not amplified, so these pin dgrad / wgrad scaling bugs that
     the train-mode norms (G5, 10-25 % bands) cannot see.
G9b  QAT at FIXED quantisation parameters: the reference QAT model after the two G9 passes, observers disabled
     (``trainer.py:299-300``), one more train-mode forward + backward on a third batch: every BatchNorm's running
     statistics and every observer's qparams BEFORE that pass (the state the HIP model is loaded with), the losses, and
     the fake-quantised outputs of five layers (sub-sampled).
G12  on-disk checkpoint layouts: ``Trainer.save`` (``trainer.py:187-202``) through a DataParallel-wrapped and a bare model,
     a backbone-prefix file cut by ``convert.partial``'s rule (``convert.py:70-78``): top-level keys / value types and the (key, shape, dtype) list of the stored
     state dicts.  Weights themselves are not stored.

Writes ``golden_r2.npz`` and ``g12_checkpoints.json`` next to this file.
"""
import hashlib
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from seedfill import seeded_boxes, seeded_fill, seeded_images  # noqa: E402

FULL_GRADS = {      # parameters whose full gradient tensor is stored (kept small: <= ~150k elements each)
    'mobilenetv2-fpn': ['module_list.0.conv.weight', 'module_list.0.bn.weight', 'module_list.5.conv.weight',
                        'module_list.25.conv.weight', 'module_list.39.conv.weight', 'module_list.39.bn.bias',
                        'module_list.49.conv.weight', 'module_list.74.conv.weight', 'module_list.92.conv.weight',
                        'module_list.101.conv.weight', 'module_list.101.conv.bias', 'module_list.71.conv.bias'],
    'regnetx-600m-fpn': ['module_list.0.conv.weight', 'module_list.0.bn.weight', 'module_list.1.conv.weight',
                         'module_list.4.conv.weight', 'module_list.10.conv.weight', 'module_list.10.bn.bias',
                         'module_list.17.conv.weight', 'module_list.85.conv.weight', 'module_list.124.conv.weight',
                         'module_list.127.conv.weight', 'module_list.127.conv.bias'],
}
QAT_LAYERS = [1, 23, 49, 71, 92]     # conv blocks whose fake-quantised output is recorded (sub-sampled to <= 40k values)


def main():
    warnings.filterwarnings('ignore')
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, interp, rloss, rparser = mg.import_reference()
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    from oracle import postprocess as opost
    npz = {}

    # ------------------------------------------------------------------ G4b ciou without NaN
    g = torch.Generator().manual_seed(44)
    xy1, wh1 = torch.rand(12, 2, generator=g) * 50, torch.rand(12, 2, generator=g) * 40 + 2
    xy2, wh2 = torch.rand(12, 2, generator=g) * 50, torch.rand(12, 2, generator=g) * 40 + 2
    b1, b2 = torch.cat([xy1, xy1 + wh1], 1), torch.cat([xy2, xy2 + wh2], 1)
    b2[3] = torch.tensor([b1[3, 0] + 1, b1[3, 1] + 1, b1[3, 2] - 1, b1[3, 3] - 1])      # contained
    b2[5] = b1[5] + torch.tensor([100., 100., 100., 100.])                               # disjoint
    npz['g4b_b1'], npz['g4b_b2'] = mg.t2n(b1), mg.t2n(b2)
    for name in ('iou_calc3', 'giou', 'diou', 'ciou'):
        v = getattr(tools, name)(b1, b2)
        assert torch.isfinite(v).all(), name
        npz['g4b_' + name] = mg.t2n(v)
        npz['g4b_bc_' + name] = mg.t2n(getattr(tools, name)(b1[:, None, :], b2[None, :, :]))
    assert np.isfinite(npz['g4b_bc_ciou']).all()

    # ------------------------------------------------------------------ G5b eval-mode BatchNorm + loss, 512x512 batch 2
    def targets(batch, size, ncls, seed):
        hw = np.array([[size // s, size // s] for s in (8, 16, 32)])
        per = [opost.create_label(bb, hw, ncls) for bb in seeded_boxes(batch, size, ncls, seed)]
        return tuple([torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
                     + [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)])

    names = {}
    for tag, cfg in (('mbv2', 'mobilenetv2-fpn'), ('rxfpn', 'regnetx-600m-fpn')):
        net = interp.DetectionModel(os.path.join(mg.REF, 'model/cfg', cfg + '.cfg'))
        seeded_fill(net, 5)
        net.eval()
        x = seeded_images(2, 512, 512, 5)
        tgt = targets(2, 512, 20, 6)
        out = net(x, tgt)
        out['loss'].mean().backward()
        npz['g5b_%s_loss' % tag] = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                                            + [v.item() for v in out['loss_per_branch']])
        params = dict(net.named_parameters())
        names[tag] = list(params)
        npz['g5b_%s_gradnorm' % tag] = np.array([p.grad.double().norm().item() for p in params.values()])
        for k in FULL_GRADS[cfg]:
            npz['g5b_%s_grad_%s' % (tag, k)] = mg.t2n(params[k].grad)
        print('G5b', cfg, npz['g5b_%s_loss' % tag])

    # ------------------------------------------------------------------ G9b QAT at fixed qparams
    import torch.ao.quantization as tq
    torch.quantization.fuse_modules = lambda m, names_, inplace=True: tq.fuse_modules_qat(m, names_, inplace=inplace)
    net = interp.DetectionModel(os.path.join(mg.REF, 'model/cfg/mobilenetv2-fpn.cfg'), quant=True)
    seeded_fill(net, 0)
    net.train()
    tools.fuse_model(net)
    tools.prepare_qat(net, backend='qnnpack')
    B, S, ncls = 2, 256, 20
    for step in range(2):                       # the two G9 passes: observers settle, BatchNorm statistics move
        net.zero_grad()
        net(seeded_images(B, S, S, step), targets(B, S, ncls, 1 + step))['loss'].mean().backward()
    net.apply(torch.quantization.disable_observer)
    sd = net.state_dict()
    bn_keys = [k for k in sd if k.endswith('running_mean') or k.endswith('running_var')]
    npz['g9b_bn_stats'] = np.concatenate([mg.t2n(sd[k]).ravel() for k in bn_keys])
    obs_names, obs = [], []
    for k, m in net.named_modules():
        if type(m).__name__ == 'FusedMovingAvgObsFakeQuantize':
            obs_names.append(k)
            ap = m.activation_post_process
            obs.append([ap.min_val.item(), ap.max_val.item(), m.scale.item(), float(m.zero_point.item())])
    npz['g9b_observers'] = np.array(obs, dtype=np.float64)
    taps = {}
    hooks = [net.module_list[i].register_forward_hook(lambda mod, inp, outp, i=i: taps.__setitem__(i, outp.detach().clone()))
             for i in QAT_LAYERS]
    net.zero_grad()
    out = net(seeded_images(B, S, S, 2), targets(B, S, ncls, 3))
    out['loss'].mean().backward()
    for h in hooks:
        h.remove()
    npz['g9b_loss'] = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                               + [v.item() for v in out['loss_per_branch']])
    for i in QAT_LAYERS:
        t = taps[i].permute(0, 2, 3, 1).reshape(-1)          # NHWC order, flattened
        stride = max(1, t.numel() // 40000)
        npz['g9b_act%d' % i] = mg.t2n(t[::stride])
        npz['g9b_act%d_stride' % i] = np.array([stride, t.numel()])
    obs2 = [[m.activation_post_process.min_val.item(), m.activation_post_process.max_val.item(), m.scale.item(),
             float(m.zero_point.item())] for k, m in net.named_modules() if type(m).__name__ == 'FusedMovingAvgObsFakeQuantize']
    assert np.array_equal(np.array(obs2), npz['g9b_observers']), 'observers moved although disabled'
    params = dict(net.named_parameters())
    npz['g9b_gradnorm'] = np.array([p.grad.double().norm().item() for p in params.values()])
    print('G9b losses', npz['g9b_loss'])

    # ------------------------------------------------------------------ G12 checkpoint layouts
    import trainer
    cfg_path = os.path.join(mg.REF, 'model/cfg/mobilenetv2-fpn.cfg')

    def describe(sd_):
        return [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd_.items()]

    ck = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, wrap in (('dp', True), ('bare', False)):
            model = interp.DetectionModel(cfg_path)
            if wrap:
                model = torch.nn.DataParallel(model)
            fake = types.SimpleNamespace(_weight_base_name='model', _weights_dir=tmp, AP=None, global_step=1234, model=model,
                                         cfg=open(cfg_path).read(), _quant_train=False, _quant_backend='qnnpack')
            trainer.Trainer.save(fake, 7)
            path = os.path.join(tmp, 'model-7.pt')
            st = torch.load(path, map_location='cpu', weights_only=False)
            ck[tag] = {'file': 'model-7.pt', 'top': {k: type(v).__name__ for k, v in st.items()}, 'step': st['step'],
                       'type': st['type'], 'backend': st['backend'], 'cfg_sha256': hashlib.sha256(st['cfg'].encode()).hexdigest(),
                       'cfg_len': len(st['cfg']), 'model': describe(st['model'])}
            if not wrap:
                # backbone file = layers 0..61: the keys of the saved dict up to the first one that names layer 62 (the rule of
                # convert.py:70-78, whose module cannot be imported here: it needs onnx)
                bb = {}
                for key, val in st['model'].items():
                    if '62.' in key:
                        break
                    bb[key] = val
                ck['backbone61'] = {'n_keys': len(bb), 'first': list(bb)[0], 'last': list(bb)[-1], 'model': describe(bb)}
            os.remove(path)
    with open(os.path.join(HERE, 'g12_checkpoints.json'), 'w') as f:
        json.dump(ck, f)

    np.savez_compressed(os.path.join(HERE, 'golden_r2.npz'), **npz)
    with open(os.path.join(HERE, 'g_r2_names.json'), 'w') as f:
        json.dump({'g5b': names, 'g9b_observers': obs_names, 'g9b_bn_keys': bn_keys, 'g9b_params': list(params)}, f)
    print('wrote', len(npz), 'arrays')


if __name__ == '__main__':
    main()
