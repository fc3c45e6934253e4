"""A driver written against the REFERENCE's import names and call pattern -- ``import tools``, ``tools.build_model``, the
loop body of ``trainer.py:225-242`` (scheduler -> forward -> zero_grad -> ``losses['loss'].mean().backward()`` -> step) and
``Trainer.save`` / resume (``trainer.py:187-202``, ``tools.py:194-203``).  It never mentions pqdet_amd: the launcher
(``python -m pqdet_amd.launch``) makes these names resolve.  Prints one JSON line per rank."""
import json
import os
import sys

import numpy as np
import torch

import tools                                   # noqa: E402  (the reference needs tools before model.*)
from model.interpreter import DetectionModel   # noqa: E402,F401
from dataset import labels                     # noqa: E402


def main():
    cfg_path, out_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank = int(os.environ.get('RANK', '0'))
    device = torch.device('cuda', torch.cuda.current_device())
    model, info = tools.build_model(cfg_path, None, None, device=device, clear_history=False, dataparallel=True,
                                    device_ids=None, qat=False, backend=None)
    global_step = info.get('step', 0)
    optimizer = torch.optim.Adam(model.parameters(), lr=4e-4, weight_decay=0.0) if os.environ.get('PQ_DRIVER_TORCH_ADAM') \
        else __import__('pqdet_amd.optim', fromlist=['Adam']).Adam(model.parameters(), lr=4e-4, weight_decay=0.0)
    B, S = 2, 128
    g = torch.Generator().manual_seed(100 + rank)
    image = torch.randn(B, 3, S, S, generator=g).to(device)
    tgt = tuple(t.to(device) for t in labels.synthetic_targets(B, S, 20, seed=7 + rank))
    model.train()
    hist = []
    for _ in range(steps):
        lr = 4e-4
        for group in optimizer.param_groups:
            group['lr'] = lr
        losses = model(image, tgt)
        optimizer.zero_grad()
        losses['loss'].mean().backward()
        optimizer.step()
        global_step += 1
        hist.append(losses['loss'].item())
    status = {'step': global_step, 'AP': None, 'model': model.state_dict(), 'cfg': open(cfg_path).read(), 'type': 'normal',
              'backend': 'none'}
    path = os.path.join(out_dir, 'model-%d.pt' % rank)
    torch.save(status, path)
    # resume exactly like Trainer.init_model: cfg may come from the checkpoint
    model2, info2 = tools.build_model(None, path, device=device, dataparallel=False)
    same = all(torch.equal(a.detach().cpu(), b.detach().cpu())
               for a, b in zip(tools.bare_model(model).state_dict().values(), tools.bare_model(model2).state_dict().values()))
    first = next(iter(tools.bare_model(model).parameters())).detach().float().cpu().double().sum().item()
    print(json.dumps({'rank': rank, 'loss': hist, 'step': info2['step'], 'resumed_equal': bool(same),
                      'keys_have_module_prefix': next(iter(status['model'])).startswith('module.'),
                      'wrapped': type(model).__name__, 'param_checksum': first}), flush=True)


if __name__ == '__main__':
    main()
