"""Deterministic weight / input protocol shared by the golden generator and the tests.

Weights are never stored in fixtures: both sides (the imported reference in ``make_golden.py`` and the
oracle / HIP model in the tests) fill an identically-keyed ``state_dict`` with this function.
"""
import numpy as np
import torch


def seeded_fill(model: torch.nn.Module, seed: int = 0) -> None:
    """Fill every floating tensor of ``model.state_dict()`` in key order from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            if not t.is_floating_point():
                continue
            shape = tuple(t.shape)
            leaf = key.split('.')[-1]
            parent = key.split('.')[-2] if '.' in key else ''
            if parent == 'bn' and leaf == 'weight':
                v = torch.rand(shape, generator=g) + 0.5
            elif parent == 'bn' and leaf == 'running_var':
                v = torch.rand(shape, generator=g) + 0.5
            elif parent == 'bn':                      # bias, running_mean
                v = torch.randn(shape, generator=g) * 0.1
            elif leaf == 'bias':
                v = torch.randn(shape, generator=g) * 0.1
            else:                                     # conv / fc weight
                fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
                v = torch.randn(shape, generator=g) * (1.0 / np.sqrt(fan_in))
            t.copy_(v.to(t.device))


def seeded_images(batch: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, h, w, generator=g)


def seeded_boxes(batch: int, size: int, num_classes: int, seed: int = 1, max_boxes: int = 4):
    """List (per image) of float32 arrays (n,6) = x1,y1,x2,y2,cls,mixw; deterministic numpy stream."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(batch):
        n = rng.randint(1, max_boxes + 1)
        rows = []
        for _ in range(n):
            w, h = rng.uniform(size / 8, size * 0.6, 2)
            x1, y1 = rng.uniform(0, size - w), rng.uniform(0, size - h)
            rows.append([x1, y1, x1 + w, y1 + h, rng.randint(0, num_classes), rng.choice([1.0, 0.6])])
        out.append(np.asarray(rows, dtype=np.float32))
    return out
