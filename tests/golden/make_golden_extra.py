#!/usr/bin/env python3
"""Additional golden fixtures recorded by IMPORTING the reference (see make_golden.py for the harness and its rules):

G10  learning-rate schedules of the trainer (cosine + multi-step with warm-up), ``trainer.py:71-117``
G11  mAP bookkeeping of the evaluator on fixed detections / labels, ``eval/evaluator.py:64-178``
     (``add_detections`` / ``add_labels`` / ``AP`` are called directly, the NMS stage is not involved)

Writes ``golden_extra.npz`` next to this file.  Only inputs and outputs are stored.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.import_reference()
    import trainer
    import eval.evaluator as rev
    npz = {}

    # ------------------------------------------------------------------ G10 LR schedules
    opt = types.SimpleNamespace(param_groups=[{'lr': 0.0}])
    fake = types.SimpleNamespace(_warmup_epochs=2, _steps_per_epoch=37, _max_epochs=30, _init_lr=4e-4, _end_lr=1e-6,
                                 _mile_stones=[10, 20, 25], _gamma=0.1, optimizer=opt)
    steps = np.array([0, 1, 36, 73, 74, 75, 200, 369, 370, 555, 739, 740, 924, 925, 1000, 1109], dtype=np.int64)
    npz['g10_steps'] = steps
    npz['g10_params'] = np.array([2, 37, 30, 4e-4, 1e-6, 0.1])
    npz['g10_milestones'] = np.array([10, 20, 25])
    npz['g10_cosine'] = np.array([trainer.Trainer.scheduler_cosine(fake, int(s)) for s in steps])
    npz['g10_step'] = np.array([trainer.Trainer.scheduler_step(fake, int(s)) for s in steps])
    assert opt.param_groups[0]['lr'] == npz['g10_step'][-1]

    # ------------------------------------------------------------------ G11 AP bookkeeping
    def scenario(tag, seed, label_classes):
        rng = np.random.RandomState(seed)
        ncls, nimg = 4, 6
        ev = rev.Evaluator.__new__(rev.Evaluator)
        ev._classes = ['c%d' % i for i in range(ncls)]
        ev.init_statics()
        for i in range(nimg):
            ngt = rng.randint(1, 6)
            xy = rng.rand(ngt, 2) * 200
            wh = rng.rand(ngt, 2) * 80 + 20
            cls = rng.randint(0, label_classes, size=(ngt, 1))
            if i < label_classes:
                cls[0, 0] = i                                        # every labelled class occurs at least once
            labels = np.concatenate([xy, xy + wh, cls], 1).astype(np.float32)
            diffs = (rng.rand(ngt) < 0.25).astype(np.float32)
            # detections: jittered copies of the labels (some duplicated), plus random false positives
            pick = rng.randint(0, ngt, size=ngt + 2)
            jit = labels[pick, :4] + rng.randn(len(pick), 4).astype(np.float32) * 6
            det_cls = labels[pick, 4:5].copy()
            flip = rng.rand(len(pick)) < 0.15
            det_cls[flip] = rng.randint(0, ncls, size=(int(flip.sum()), 1))
            fp_xy = rng.rand(3, 2) * 250
            fp = np.concatenate([fp_xy, fp_xy + rng.rand(3, 2) * 60 + 10], 1)
            boxes = np.concatenate([jit, fp], 0)
            dcls = np.concatenate([det_cls, rng.randint(0, ncls, size=(3, 1))], 0)
            score = rng.rand(len(boxes), 1)
            dets = np.concatenate([boxes, score, dcls], 1).astype(np.float32)
            if i == 4:
                dets = dets[:0]                                      # an image without detections
            name = 'img%d' % i
            ev.add_detections(name, dets)
            ev.add_labels(name, labels, diffs)
            npz['g11%s_dets%d' % (tag, i)] = dets
            npz['g11%s_labels%d' % (tag, i)] = labels
            npz['g11%s_diffs%d' % (tag, i)] = diffs
        m = ev.AP()
        npz['g11%s_raw' % tag] = np.asarray(m.raw)
        npz['g11%s_mAPs' % tag] = np.asarray(m.mAPs)
        npz['g11%s_APs' % tag] = np.asarray(m.APs)
        npz['g11%s_AP' % tag] = np.asarray(m.AP)
        npz['g11%s_iou_thresholds' % tag] = np.asarray(m.iou_thresholds)
        return float(m.AP)

    ap_a = scenario('a', 11, 4)      # every class labelled
    ap_b = scenario('b', 12, 3)      # class 3 is detected but never labelled: the reference divides 0/0 -> NaN rows
    print('AP a = %.6f, AP b = %r' % (ap_a, ap_b))

    np.savez_compressed(os.path.join(HERE, 'golden_extra.npz'), **npz)
    print('wrote', len(npz), 'arrays')


if __name__ == '__main__':
    main()
