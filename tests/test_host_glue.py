"""Host-side glue around the hot path against fixtures recorded from the reference: LR schedules (G10), evaluator mAP
bookkeeping (G11), the in-repo NumPy NMS (G7).  No GPU needed."""
import types

import numpy as np
import pytest

from pqdet_amd import schedule, tools
from pqdet_amd.eval import Evaluator, average_precision


def _sched(kind, g):
    warm, spe, max_ep, init_lr, end_lr, gamma = g['g10_params']
    opt = types.SimpleNamespace(param_groups=[{'lr': 0.0}, {'lr': 1.0}])
    s = schedule.LRScheduler(opt, kind, float(init_lr), float(end_lr), int(warm), int(max_ep), int(spe),
                             milestones=[int(m) for m in g['g10_milestones']], gamma=float(gamma))
    return s, opt


def test_lr_schedules_match_reference(golden_extra):
    g = golden_extra
    for kind, key in (('cosine', 'g10_cosine'), ('step', 'g10_step')):
        s, opt = _sched(kind, g)
        got = np.array([s(int(st)) for st in g['g10_steps']])
        np.testing.assert_array_equal(got, g[key])                  # same double-precision expression -> bit equal
        assert all(grp['lr'] == got[-1] for grp in opt.param_groups)
    assert schedule.cosine_lr(0, 4e-4, 1e-6, 74, 1110) == 0.0


def test_evaluator_ap_bookkeeping_matches_reference(golden_extra):
    g = golden_extra
    for tag in 'ab':
        ev = Evaluator(None, classes=['c%d' % i for i in range(4)])
        for i in range(6):
            ev.add_detections('img%d' % i, g['g11%s_dets%d' % (tag, i)])
            ev.add_labels('img%d' % i, g['g11%s_labels%d' % (tag, i)], g['g11%s_diffs%d' % (tag, i)])
        m = ev.AP()
        np.testing.assert_array_equal(m.raw, g['g11%s_raw' % tag])          # NaN rows (class never labelled) included
        np.testing.assert_array_equal(m.mAPs, g['g11%s_mAPs' % tag])
        np.testing.assert_array_equal(m.APs, g['g11%s_APs' % tag])
        np.testing.assert_array_equal(np.asarray(m.AP), g['g11%s_AP' % tag])
        np.testing.assert_array_equal(m.iou_thresholds, g['g11%s_iou_thresholds' % tag])
        assert ev.detections_count == 0 and not ev.labels                    # statistics are reset after AP()
    assert Evaluator(None, classes=['a']).AP() is None                       # nothing added -> the reference returns None too


def test_average_precision_envelope():
    rec = np.array([[0.1, 0.2, 0.2, 0.5]])
    prec = np.array([[1.0, 0.5, 0.66, 0.4]])
    # envelope 1.0, 0.66, 0.66, 0.4 over recall steps 0.1, 0.1, 0, 0.3 (+ 0 for the tail to recall 1)
    np.testing.assert_allclose(average_precision(rec, prec), [0.1 * 1.0 + 0.1 * 0.66 + 0.3 * 0.4])


def test_numpy_nms_matches_reference(golden):
    for method, key in (('nms', 'g7_nms'), ('soft-nms', 'g7_softnms')):
        out = tools.nms(golden['g7_in'].copy(), 0.1, 0.45, method=method)
        np.testing.assert_array_equal(out, golden[key])
    one = np.array([[0, 0, 10, 10, 0.05, 2]], dtype=np.float32)
    np.testing.assert_array_equal(tools.nms(one, 0.1, 0.45), one)             # a class's best box is reported regardless


def test_qat_checkpoint_layout_matches_reference_and_round_trips():
    """qat_state_dict() emits exactly the reference QAT model's state_dict (names, shapes, dtypes recorded in G9) and
    load_qat_state_dict() takes it back (CPU only: no kernel is involved)."""
    import io
    import json
    import os

    import torch
    from pqdet_amd.model import zoo
    from pqdet_amd.model.interpreter import DetectionModel
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    want = json.load(open(os.path.join(here, 'g9_names.json')))['state_dict']
    net = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')), quant=True)
    tools.prepare_qat(tools.fuse_model(net))
    net._qat_state[:, 0] = torch.arange(181.) * -0.01
    net._qat_state[:, 1] = torch.arange(181.) * 0.02 + 1
    net._qat_state[:, 2] = 0.05
    net._qat_state[:, 3] = 7
    sd = tools.qat_state_dict(net)
    assert sorted(sd) == sorted(k for k, _, _ in want)
    for k, shape, dtype in want:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k
    other = DetectionModel(io.StringIO(zoo.cfg_text('mobilenetv2-fpn')), quant=True)
    tools.prepare_qat(other)
    tools.load_qat_state_dict(other, {'module.' + k: v for k, v in sd.items()})
    assert torch.equal(other._qat_state, net._qat_state)
    for (k1, v1), (_, v2) in zip(net.state_dict().items(), other.state_dict().items()):
        assert torch.equal(v1, v2), k1


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_vectorised_create_label_keeps_the_boxes_dtype_like_the_reference_loop(dtype):
    """dataset/train_dataset.py:109-150 computes centres, `// strides` and the anchor IoUs in whatever dtype the sample getter
    delivers.  Boxes whose centres sit a float32 ulp away from a cell boundary must land in the same cells as the per-box
    loop (oracle.postprocess.create_label, pinned by G8) -- for float32 AND float64 input."""
    from oracle import postprocess as opost
    from pqdet_amd.dataset import labels
    rng = np.random.RandomState(11)
    hw = np.array([[64, 64], [32, 32], [16, 16]])
    rows = []
    for k in range(64):
        cx, cy = rng.randint(1, 15, 2) * 32.0                 # centres on cell boundaries of every scale ...
        eps = (rng.rand(2) - 0.5) * 4e-5                      # ... nudged by less than a float32 ulp of the coordinate
        w, h = rng.uniform(8, 200, 2)
        x1, y1 = cx + eps[0] - w / 2, cy + eps[1] - h / 2
        rows.append([x1, y1, x1 + w, y1 + h, rng.randint(0, 20), rng.uniform(0.5, 1.0)])
    boxes = [np.array(r, dtype=dtype) for r in rows]
    want = opost.create_label(boxes, hw, 20)
    got = labels.create_label(boxes, hw, 20)
    for i in range(3):
        np.testing.assert_array_equal(got[i], want[i])
        np.testing.assert_array_equal(np.asarray(got[3 + i], np.float32).reshape(-1, 4), np.asarray(want[3 + i], np.float32).reshape(-1, 4))
