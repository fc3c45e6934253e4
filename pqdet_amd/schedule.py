"""Learning-rate schedules of the training loop (reference ``trainer.py:71-117``): linear warm-up from 0 to ``init_lr``
over ``warmup_epochs * steps_per_epoch`` steps, then either cosine annealing to ``end_lr`` at the last step or a
multi-step decay by ``gamma`` at epoch milestones.  The value is written into every ``param_group['lr']`` each step
(``trainer.py:225``); :class:`pqdet_amd.optim.Adam` reads it on the host and passes it to ``pq_adam_step`` as a scalar, so
changing it costs nothing on the device."""
import math
from itertools import chain
from typing import Iterable, Optional


def cosine_lr(step: int, init_lr: float, end_lr: float, warmup_steps: int, max_steps: int) -> float:
    if step < warmup_steps:
        return step / warmup_steps * init_lr
    phase = (step - warmup_steps) / (max_steps - warmup_steps)
    return end_lr + 0.5 * (init_lr - end_lr) * (1 + math.cos(phase * math.pi))


def step_lr(step: int, init_lr: float, warmup_steps: int, steps_per_epoch: int, milestones: Iterable[int],
            max_epochs: int, gamma: float) -> Optional[float]:
    if step < warmup_steps:
        return step / warmup_steps * init_lr
    for i, m in enumerate(chain(milestones, [max_epochs])):
        if step < m * steps_per_epoch:
            return init_lr * gamma ** i
    return None          # past the last epoch: the reference keeps the previous value


class LRScheduler:
    """``lr = sched(global_step)`` sets the optimiser's learning rate and returns it (``Trainer.scheduler``)."""

    def __init__(self, optimizer, kind: str, init_lr: float, end_lr: float, warmup_epochs: int, max_epochs: int,
                 steps_per_epoch: int, milestones=(), gamma: float = 0.1):
        if kind not in ('cosine', 'step'):
            raise KeyError(kind)
        self.optimizer, self.kind = optimizer, kind
        self.init_lr, self.end_lr = init_lr, end_lr
        self.warmup_steps = warmup_epochs * steps_per_epoch
        self.max_steps = max_epochs * steps_per_epoch
        self.steps_per_epoch, self.max_epochs = steps_per_epoch, max_epochs
        self.milestones, self.gamma = list(milestones), gamma
        self._last = 0.0

    def __call__(self, step: int) -> float:
        if self.kind == 'cosine':
            lr = cosine_lr(step, self.init_lr, self.end_lr, self.warmup_steps, self.max_steps)
        else:
            lr = step_lr(step, self.init_lr, self.warmup_steps, self.steps_per_epoch, self.milestones, self.max_epochs,
                         self.gamma)
            if lr is None:
                lr = self._last
        self._last = lr
        for group in self.optimizer.param_groups:
            group['lr'] = lr
        return lr
