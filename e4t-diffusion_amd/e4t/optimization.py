"""Learning-rate schedules of the reference's ``--lr_scheduler`` flag (pretrain_e4t.py:110-111,402-408 →
``diffusers.optimization.get_scheduler``; [3P] diffusers 0.14, restated): a multiplier lambda(step) on the base rate, stepped
once per optimiser step.  The native trainer has no ``torch.optim`` object: ``LRSchedule.apply(trainer)`` sets the scalar the
fused AdamW kernel receives."""
from __future__ import annotations

import math

SCHEDULES = ("linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup")


def get_lr_lambda(name: str, num_warmup_steps: int = 0, num_training_steps: int = 0, num_cycles=None, power: float = 1.0,
                  lr_init: float = 1.0, lr_end: float = 1e-7):
    if name not in SCHEDULES:
        raise ValueError(f"{name} is not a valid scheduler: choose from {list(SCHEDULES)}")
    w, T = num_warmup_steps, num_training_steps
    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: float(step) / float(max(1.0, w)) if step < w else 1.0
    if name == "linear":
        return lambda step: float(step) / float(max(1, w)) if step < w else max(0.0, float(T - step) / float(max(1, T - w)))
    if name == "cosine":
        nc = 0.5 if num_cycles is None else num_cycles

        def f(step):
            if step < w:
                return float(step) / float(max(1, w))
            progress = float(step - w) / float(max(1, T - w))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(nc) * 2.0 * progress)))
        return f
    if name == "cosine_with_restarts":
        nc = 1 if num_cycles is None else num_cycles

        def f(step):
            if step < w:
                return float(step) / float(max(1, w))
            progress = float(step - w) / float(max(1, T - w))
            if progress >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(nc) * progress) % 1.0))))
        return f
    # polynomial
    if not lr_init > lr_end:
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def f(step):
        if step < w:
            return float(step) / float(max(1, w))
        if step > T:
            return lr_end / lr_init
        decay = (lr_init - lr_end) * (1 - (step - w) / (T - w)) ** power + lr_end
        return decay / lr_init
    return f


class LRSchedule:
    """``get_scheduler(name, optimizer, num_warmup_steps, num_training_steps)`` for the native trainer."""

    def __init__(self, name, base_lr, num_warmup_steps=0, num_training_steps=0, last_step=0):
        self.base_lr, self.step_count = base_lr, last_step
        self.fn = get_lr_lambda(name, num_warmup_steps, num_training_steps, lr_init=base_lr)

    def get_last_lr(self):
        return [self.base_lr * self.fn(self.step_count)]

    def apply(self, trainer):
        trainer.lr = self.get_last_lr()[0]

    def step(self):
        self.step_count += 1
