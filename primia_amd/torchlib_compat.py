"""Host-side mirror of the pieces of PriMIA's `torchlib/utils.py` that sit on the hot path
(SURVEY.md §8b): same names, arguments and error behaviour, running on the HIP engine.

This file holds control flow only (schedules, sync bookkeeping, argument plumbing); every tensor
operation is a HIP kernel behind primia_amd.engine / primia_amd.fed.
"""
from typing import Optional

import numpy as np


class LearningRateScheduler:
    """torchlib/utils.py:37-89.

    Available schedule plans:
    log_linear : linear interpolation of log10(lr)
    log_cosine : cosine interpolation of log10(lr)
    `restarts` splits the run into restarts+1 equal cycles.
    """

    def __init__(self, total_epochs: int, log_start_lr: float, log_end_lr: float,
                 schedule_plan: str = "log_linear", restarts: Optional[int] = None):
        if restarts == 0:
            restarts = None
        self.total_epochs = total_epochs if not restarts else total_epochs / (restarts + 1)
        span = log_end_lr - log_start_lr
        if schedule_plan == "log_linear":
            self.calc_lr = lambda epoch: np.power(10, (span / self.total_epochs) * epoch + log_start_lr)
        elif schedule_plan == "log_cosine":
            self.calc_lr = lambda epoch: np.power(
                10, (np.cos(np.pi * (epoch / self.total_epochs)) / 2.0 + 0.5) * abs(span) + log_end_lr)
        else:
            raise NotImplementedError(
                "Requested learning rate schedule {} not implemented".format(schedule_plan))

    def get_lr(self, epoch: int):
        epoch = epoch % self.total_epochs
        if (type(epoch) is int and epoch > self.total_epochs) or (
                type(epoch) is np.ndarray and np.max(epoch) > self.total_epochs):
            raise AssertionError("Requested epoch out of precalculated schedule")
        return self.calc_lr(epoch)

    def adjust_learning_rate(self, optimizer, epoch: int):
        """`optimizer` is anything with `param_groups` (torch optimizer) or an `lr` attribute."""
        new_lr = self.get_lr(epoch)
        if hasattr(optimizer, "param_groups"):
            for param_group in optimizer.param_groups:
                param_group["lr"] = new_lr
        else:
            optimizer.lr = new_lr
        return new_lr
