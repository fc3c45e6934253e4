"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the PQDet detection hot path.

Nothing under ``pqdet_amd/`` imports this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker / the reported CPU baseline.

Parity status: PINNED for everything the reference implements in-repo -- the restatement is checked against
golden vectors produced by importing the reference itself (``tests/golden/*.npz|json``, generator
``tests/golden/make_golden.py``).  UNPINNED at one third-party boundary: ``tools.torch_nms`` delegates its
suppression arithmetic to ``torchvision.ops.boxes.batched_nms`` (torchvision pinned ``==0.6.0`` in the
reference's ``requirements.txt:3``; not vendored, not installed here).  ``oracle/postprocess.py`` restates the
published torchvision algorithm (class offset ``cls * (max_coord + 1)``, IoU without +1, suppress iff
``iou > thr``, score-descending order, ties by ascending candidate index); see DESIGN.md.
"""
