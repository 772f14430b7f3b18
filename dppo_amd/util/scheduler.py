"""Cosine learning-rate schedule with linear warm-up and restarts.

Same behaviour as the reference's ``CosineAnnealingWarmupRestarts`` (dppo/util/scheduler.py:32-147) on the
``step()``-without-epoch path the agents use; written against anything that exposes ``param_groups`` so it
drives both torch optimisers and dppo_amd's FlatAdamW.
"""
import math


class CosineAnnealingWarmupRestarts:
    def __init__(self, optimizer, first_cycle_steps, cycle_mult=1.0, max_lr=0.1, min_lr=0.001, warmup_steps=0,
                 gamma=1.0, last_epoch=-1):
        assert warmup_steps < first_cycle_steps
        self.optimizer = optimizer
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.base_max_lr = self.max_lr = max_lr
        self.min_lr, self.warmup_steps, self.gamma = min_lr, warmup_steps, gamma
        self.cur_cycle_steps = first_cycle_steps
        self.cycle = 0
        self.step_in_cycle = last_epoch
        self.last_epoch = last_epoch
        self.base_lrs = [min_lr for _ in optimizer.param_groups]
        self.step()  # the torch base class takes one initial step (step_in_cycle 0) ...
        for group in optimizer.param_groups:  # ... and the reference's init_lr() (:76-80) then sets min_lr, whatever that
            group["lr"] = min_lr  # step computed (with warmup_steps = 0 it computed max_lr)

    def get_lr(self):
        if self.step_in_cycle == -1:
            return list(self.base_lrs)
        if self.step_in_cycle < self.warmup_steps:
            return [(self.max_lr - b) * self.step_in_cycle / self.warmup_steps + b for b in self.base_lrs]
        phase = math.pi * (self.step_in_cycle - self.warmup_steps) / (self.cur_cycle_steps - self.warmup_steps)
        return [b + (self.max_lr - b) * (1 + math.cos(phase)) / 2 for b in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        self.step_in_cycle += 1
        if self.step_in_cycle >= self.cur_cycle_steps:
            self.cycle += 1
            self.step_in_cycle -= self.cur_cycle_steps
            self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group["lr"] = lr
