"""Debug helper: per-parameter relative gradient error of the HIP engine vs the CPU oracle (run on the GPU box)."""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from seedfill import seeded_boxes, seeded_fill, seeded_images
from oracle import postprocess as opost
from oracle.net import OracleNet
from pqdet_amd.model import zoo
from pqdet_amd.model.interpreter import DetectionModel

def targets(batch, h, w, ncls, seed=1):
    bl = seeded_boxes(batch, min(h, w), ncls, seed)
    hw = np.array([[h // s, w // s] for s in (8, 16, 32)])
    per = [opost.create_label(bb, hw, ncls) for bb in bl]
    labels = [torch.from_numpy(np.stack([p[i] for p in per])) for i in range(3)]
    boxes = [torch.from_numpy(opost.pad_boxes([p[3 + i] for p in per])) for i in range(3)]
    return tuple(labels + boxes)

