#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference (read-only, from
``/root/reference``) and recording inputs + outputs.  No reference source text is stored.

Run once in the build container:  ``python tests/golden/make_golden.py``
The reference cannot travel to the GPU box; the tests only read the ``.npz`` / ``.json`` written here.

Third-party modules the reference imports but that are absent here (cv2, torchvision, yacs) are replaced by
empty stand-in *modules* so that ``import tools`` succeeds; none of their functions are called for the
fixtures below (``torch_nms``'s torchvision call is therefore NOT captured -- see oracle/__init__.py).
"""
import glob
import io
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('PQDET_REFERENCE', '/root/reference')
sys.path.insert(0, HERE)
from seedfill import seeded_boxes, seeded_fill, seeded_images  # noqa: E402


def import_reference():
    for m in ['cv2', 'torchvision', 'torchvision.ops', 'torchvision.ops.boxes']:
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules['torchvision'].ops = sys.modules['torchvision.ops']
    sys.modules['torchvision.ops'].boxes = sys.modules['torchvision.ops.boxes']
    yacs = types.ModuleType('yacs')
    yacs_config = types.ModuleType('yacs.config')

    class CfgNode(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules.setdefault('yacs', yacs)
    sys.modules.setdefault('yacs.config', yacs_config)
    for alias, typ in (('float', float), ('bool', bool), ('int', int)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    sys.path.insert(0, REF)
    import tools  # noqa: F401  (must precede model.* : circular import in the reference)
    import model.interpreter as interp
    import model.loss as rloss
    import model.parser as rparser
    return tools, interp, rloss, rparser


def t2n(t):
    return t.detach().cpu().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tools, interp, rloss, rparser = import_reference()

    # ---------------------------------------------------------------- G1 parser structure, all cfgs
    g1 = {}
    cfgs = sorted(glob.glob(os.path.join(REF, 'model/cfg/**/*.cfg'), recursive=True))
    for path in cfgs:
        rel = os.path.relpath(path, os.path.join(REF, 'model/cfg'))
        with open(path) as f:
            specs = list(rparser.Parser(f).gen_layers())
        with open(path) as f:
            mods = rparser.Parser(f).torch_layers()
        layers = []
        for m in mods:
            d = {'type': m._type, 'stride': m._stride, 'out': m._output_channels,
                 'notprune': bool(getattr(m, '_notprune', False))}
            if m._type == 'convolutional':
                c = m.conv
                d['conv'] = [c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0], c.padding[0],
                             c.groups, c.bias is not None, hasattr(m, 'bn'),
                             type(m.act).__name__ if hasattr(m, 'act') else 'none']
            if hasattr(m, '_from'):
                d['from'] = m._from
            if hasattr(m, '_layers'):
                d['layers'] = list(m._layers)
            if m._type == 'yolo':
                d['opt'] = {k: v for k, v in m.opt.items()}
            layers.append(d)
        nparam = sum(p.numel() for m in mods for p in m.parameters())
        g1[rel] = {'n_sections': len(specs), 'n_layers': len(mods), 'params': nparam, 'layers': layers,
                   'specs': specs}
    # lexer corner cases (synthetic text -> parsed value or error string)
    corner = {}
    snippets = {
        'ws_in_list': '[route]\nlayers = -1, 46\n',
        'comment_tail': '[yolo]\nignore_thresh = .5   # keep\nbbox_loss=l1 #x\n',
        'trailing_comma': '[route]\nlayers = -1,\n',
        'empty_mid': '[route]\nlayers = 1,,2\n',
        'sci_is_string': '[yolo]\nl1_loss_gain=1e-3\n',
        'attr_before_section': 'filters=3\n[convolutional]\nfilters=8\n',
        'no_equals': '[convolutional]\nfilters\n',
        'missing_value': '[convolutional]\nfilters=\n',
        'bad_start': '[convolutional]\n3x=1\n',
        'open_only': '[\n',
        'unknown_section': '[banana]\nx=1\n',
        'empty_section': '[]\n',
        'hash_value': '[convolutional]\nfilters = #none\n',
        'spaced_key': '[convolutional]\n  filters   =   16  \n',
        'digits_in_key': '[yolo]\nl1_loss_gain = 0.25\n',
        'no_close_bracket': '[convolutional\nfilters=4\n',
        'float_list': '[yolo]\nfoo = 1.5, 2, bar\n',
    }
    for name, text in snippets.items():
        try:
            corner[name] = {'ok': list(rparser.Parser(io.StringIO(text)).gen_layers())}
        except Exception as e:  # noqa: BLE001
            corner[name] = {'err': type(e).__name__, 'msg': str(e)}
    with open(os.path.join(HERE, 'g1_parser.json'), 'w') as f:
        json.dump({'cfgs': g1, 'corner': corner, 'snippets': snippets}, f, sort_keys=True)

    npz = {}
    # ---------------------------------------------------------------- G2 decode
    g = torch.Generator().manual_seed(2)
    for tag, shape, ncls, stride in (('a', (2, 75, 4, 4), 20, 32), ('b', (1, 45, 2, 3), 10, 8)):
        x = torch.randn(shape, generator=g)
        dec = rparser.Decode(ncls, stride)
        npz['g2_%s_in' % tag] = t2n(x)
        npz['g2_%s_out' % tag] = t2n(dec(x))
        npz['g2_%s_meta' % tag] = np.array([ncls, stride])

    # ---------------------------------------------------------------- G3 loss fwd + grad, 5 box losses
    B, H, W, A, C, M = 2, 4, 4, 3, 3, 3
    stride = 16
    g = torch.Generator().manual_seed(3)
    raw = torch.randn(B, A * (5 + C), H, W, generator=g) * 0.7
    label = torch.zeros(B, H, W, A, 6 + C)
    label[..., -1] = 1.0
    boxes = torch.zeros(B, M, 4)
    gt = [[(5., 9., 40., 50., 0, 1.0), (20., 4., 60., 33., 2, 0.6)], [(1., 1., 30., 62., 1, 1.0)]]
    for b, lst in enumerate(gt):
        for j, (x1, y1, x2, y2, c, mw) in enumerate(lst):
            cx, cy = int((x1 + x2) / 2 // stride), int((y1 + y2) / 2 // stride)
            onehot = torch.full((C,), 0.01 / C)
            onehot[c] += 0.99
            label[b, cy, cx, j % A] = torch.tensor([x1, y1, x2, y2, 1.0, *onehot.tolist(), mw])
            boxes[b, j] = torch.tensor([x1, y1, x2, y2])
    # saturate one class prob (BCE clamp) and plant an ignore-mask hit (pred ~= a gt box on a background cell)
    raw[0, 5 + 1, 0, 0] = 120.0
    raw[1, 0:4, 2, 2] = torch.log(torch.tensor([40. / 16 - 1. / 16, 40. / 16 - 1. / 16,
                                                30. / 16 - 40. / 16 + 2.5, 62. / 16 - 40. / 16 + 1.4]).clamp_min(1e-3))
    npz['g3_raw'], npz['g3_label'], npz['g3_boxes'] = t2n(raw), t2n(label), t2n(boxes)
    for kind in ('l1', 'giou', 'diou', 'ciou', 'iou'):
        opt = {'stride': stride, 'bbox_loss': kind, 'ignore_thresh': 0.5, 'l1_loss_gain': 0.05, 'classes': C}
        r = raw.clone().requires_grad_(True)
        pred = rparser.Decode(C, stride)(r)
        pred.retain_grad()
        try:
            out = rloss.loss_per_scale(pred, label, boxes, opt)
        except RuntimeError as e:
            # as written, ciou evaluates atan(0/0) on every non-responsible cell -> NaN -> the reference raises
            npz['g3_%s_raises' % kind] = np.array([1])
            print('G3', kind, 'raises:', e)
            continue
        out[0].sum().backward()
        npz['g3_%s_losses' % kind] = np.concatenate([t2n(o) for o in out])
        npz['g3_%s_dpred' % kind] = t2n(pred.grad)
        npz['g3_%s_draw' % kind] = t2n(r.grad)
    npz['g3_pred'] = t2n(rparser.Decode(C, stride)(raw))

    # ---------------------------------------------------------------- G4 IoU family
    b1 = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.], [2., 2., 8., 8.], [0., 0., 10., 10.],
                       [1., 2., 7., 9.], [3., 3., 4., 9.]])
    b2 = torch.tensor([[20., 20., 30., 30.], [0., 0., 10., 10.], [0., 0., 10., 10.], [5., 5., 15., 15.],
                       [2., 1., 11., 6.], [0., 1., 8., 3.]])
    npz['g4_b1'], npz['g4_b2'] = t2n(b1), t2n(b2)
    for name in ('iou_calc3', 'giou', 'diou', 'ciou'):
        npz['g4_' + name] = t2n(getattr(tools, name)(b1, b2))
        npz['g4_bc_' + name] = t2n(getattr(tools, name)(b1[:, None, :], b2[None, :, :]))
    npz['g4_iou_calc1'] = tools.iou_calc1(t2n(b1), t2n(b2))
    xywh1 = np.array([30., 40., 20., 10.], np.float32)
    xywh2 = np.array([[28., 44., 10., 13.], [36., 36., 33., 23.], [100., 100., 5., 5.]], np.float32)
    npz['g4_xywh1'], npz['g4_xywh2'] = xywh1, xywh2
    npz['g4_iou_xywh'] = tools.iou_xywh_numpy(xywh1, xywh2)

    # ---------------------------------------------------------------- G8 create_label (+ collate padding)
    import dataset.train_dataset as td

    class FakeDS:
        _gt_per_grid = 3
        _strides = np.array([8, 16, 32])
        _anchors = np.array([(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90),
                             (156, 198), (373, 326)], dtype=np.float32)
        _anchors_iou_threshold = 0.3

        def __init__(self, ncls):
            self._num_classes = ncls
    size = 128
    out_sizes = np.array([size, size]) // FakeDS._strides[:, None]
    lab_boxes = np.array([[10, 12, 60, 70, 1, 1.0], [14, 10, 58, 74, 3, 0.6], [90, 90, 97, 99, 0, 1.0],
                          [0, 0, 127, 127, 4, 1.0], [64, 64, 66, 120, 2, 1.0]], dtype=np.float32)
    lab = td.TrainDataset.create_label(FakeDS(5), lab_boxes, out_sizes)
    npz['g8_boxes'] = lab_boxes
    for i in range(3):
        npz['g8_label%d' % i] = lab[i]
        npz['g8_list%d' % i] = np.asarray(lab[3 + i], dtype=np.float32).reshape(-1, 4)
    padded = td._pad_arrays([np.zeros((0, 4), np.float32)[:0].tolist(), lab[3], lab[4]])
    npz['g8_padded'] = np.stack([np.asarray(p, dtype=np.float32) for p in padded])

    # ---------------------------------------------------------------- G5 full model, seeded fill
    def make_targets(batch, size, ncls, seed):
        bl = seeded_boxes(batch, size, ncls, seed)
        hw = np.array([size, size]) // FakeDS._strides[:, None]
        per = [td.TrainDataset.create_label(FakeDS(ncls), bb, hw) for bb in bl]
        labels = [torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
        lists = [td._pad_arrays([np.asarray(p[3 + i], np.float32).reshape(-1, 4) for p in per]) for i in range(3)]
        return labels + [torch.from_numpy(np.stack(l)) for l in lists]

    def run_model(tag, cfg_rel, batch, size, store_full):
        path = os.path.join(REF, 'model/cfg', cfg_rel)
        torch.manual_seed(0)
        net = interp.DetectionModel(path)
        seeded_fill(net, 0)
        x = seeded_images(batch, size, size, 0)
        ncls = [m.opt['classes'] for m in net.module_list if m._type == 'yolo'][0]
        tgt = make_targets(batch, size, ncls, 1)
        net.eval()
        with torch.no_grad():
            ev = net(x)
        net.train()
        out = net(x, tgt)
        out['loss'].mean().backward()
        if store_full:
            npz['g5_%s_eval' % tag] = t2n(ev)
        else:
            npz['g5_%s_eval_head' % tag] = t2n(ev[:, :64])
            npz['g5_%s_eval_sum' % tag] = np.array([ev.double().sum().item(), ev.double().abs().sum().item()])
        npz['g5_%s_loss' % tag] = np.array([out[k].item() for k in ('loss', 'giou_loss', 'conf_loss', 'class_loss')]
                                           + [v.item() for v in out['loss_per_branch']])
        names, gn = [], []
        for k, p in net.named_parameters():
            names.append(k)
            gn.append(p.grad.double().norm().item())
        npz['g5_%s_gradnorm' % tag] = np.array(gn)
        sd = net.state_dict()
        rm = [k for k in sd if k.endswith('running_mean')]
        npz['g5_%s_rmean_first' % tag] = t2n(sd[rm[0]])
        npz['g5_%s_rvar_first' % tag] = t2n(sd[rm[0].replace('running_mean', 'running_var')])
        npz['g5_%s_rmean_last' % tag] = t2n(sd[rm[-1]])
        npz['g5_%s_rvar_last' % tag] = t2n(sd[rm[-1].replace('running_mean', 'running_var')])
        first_w = [k for k in names if k.endswith('conv.weight')][0]
        npz['g5_%s_grad_first_w' % tag] = t2n(dict(net.named_parameters())[first_w].grad)
        last_b = [k for k in names if k.endswith('conv.bias')][-1]
        npz['g5_%s_grad_last_b' % tag] = t2n(dict(net.named_parameters())[last_b].grad)
        return names

    meta = {}
    meta['mbv2_64'] = run_model('mbv2_64', 'mobilenetv2-fpn.cfg', 2, 64, True)
    meta['rxfpn_64'] = run_model('rxfpn_64', 'regnetx-600m-fpn.cfg', 2, 64, True)
    meta['rxpan_64'] = run_model('rxpan_64', 'regnetx-600m-pan.cfg', 2, 64, True)
    meta['mbv2_512'] = run_model('mbv2_512', 'mobilenetv2-fpn.cfg', 2, 512, False)      # BASELINE config 1 shape

    # ---------------------------------------------------------------- G6 recover (voc + visdrone)
    import dataset.base_sample  # noqa: F401
    import dataset.visdrone_sample as vis
    import dataset.voc_sample as voc
    g = torch.Generator().manual_seed(6)
    pred = torch.rand(2, 7, 5 + 3, generator=g)
    pred[..., :4] = pred[..., :4] * 600 - 40
    npz['g6_pred'] = t2n(pred)
    insz = torch.FloatTensor([512, 512])
    orig = torch.FloatTensor([[375, 500], [333, 480]])
    npz['g6_orig'] = t2n(orig)
    npz['g6_voc'] = t2n(voc.recover_bboxes_prediction_voc(pred.clone(), insz, orig.clone()))
    npz['g6_voc_single'] = t2n(voc.recover_bboxes_prediction_voc(pred.clone(), insz, orig[0].clone()))
    npz['g6_visdrone'] = t2n(vis.recover_bboxes_prediction_visdrone(pred.clone(), insz, orig.clone()))

    # ---------------------------------------------------------------- G7 NMS (in-repo parts only)
    g = torch.Generator().manual_seed(7)
    n = 40
    xy = torch.rand(n, 2, generator=g) * 80
    wh = torch.rand(n, 2, generator=g) * 40 + 5
    sc = torch.rand(n, generator=g)           # tie-free with overwhelming probability
    cl = torch.randint(0, 3, (n,), generator=g).float()
    nb = torch.cat([xy, xy + wh, sc[:, None], cl[:, None]], 1).numpy().astype(np.float32)
    npz['g7_in'] = nb
    npz['g7_nms'] = tools.nms(nb.copy(), 0.1, 0.45)
    npz['g7_softnms'] = tools.nms(nb.copy(), 0.1, 0.45, method='soft-nms')
    # pre-stage of torch_nms (tools.py:550-555) on a multi-label score matrix
    ml = torch.rand(12, 4 + 5, generator=g)
    ml[:, :4] = torch.cat([ml[:, :2] * 50, ml[:, :2] * 50 + ml[:, 2:4] * 30 + 1], 1)
    mask = ml[:, 4:] > 0.4
    idx = mask.nonzero()
    npz['g7_ml_in'] = t2n(ml)
    npz['g7_ml_scores'] = t2n(ml[:, 4:][mask])
    npz['g7_ml_cls'] = t2n(idx[:, 1])
    npz['g7_ml_boxes'] = t2n(ml[:, :4][idx[:, 0]])

    np.savez_compressed(os.path.join(HERE, 'golden.npz'), **npz)
    with open(os.path.join(HERE, 'g5_param_names.json'), 'w') as f:
        json.dump(meta, f)
    print('wrote', len(npz), 'arrays')


if __name__ == '__main__':
    main()
