"""Learning-rate schedules of the training script (examples/brushnet/train_brushnet_mirror.py:1257-1263:
`diffusers.optimization.get_scheduler(args.lr_scheduler, optimizer, num_warmup_steps, num_training_steps, num_cycles)`).

Same names, arguments and values as src/diffusers/optimization.py:30-362 (checked against the reference's own schedulers in
tests/golden/lr_schedules.json), over this package's `training.AdamW` instead of a torch optimizer: the schedule is a
multiplier of the optimizer's initial learning rate, evaluated on the host and written to `optimizer.lr` (mf_adamw takes the
learning rate as a launch argument, so nothing on the device changes).
"""
import math
from enum import Enum
from typing import Callable, Optional, Union

__all__ = ["SchedulerType", "LambdaLR", "get_scheduler", "get_constant_schedule", "get_constant_schedule_with_warmup",
           "get_piecewise_constant_schedule", "get_linear_schedule_with_warmup", "get_cosine_schedule_with_warmup",
           "get_cosine_with_hard_restarts_schedule_with_warmup", "get_polynomial_decay_schedule_with_warmup"]


class SchedulerType(Enum):
    LINEAR = "linear"
    COSINE = "cosine"
    COSINE_WITH_RESTARTS = "cosine_with_restarts"
    POLYNOMIAL = "polynomial"
    CONSTANT = "constant"
    CONSTANT_WITH_WARMUP = "constant_with_warmup"
    PIECEWISE_CONSTANT = "piecewise_constant"


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR's contract for one parameter group: lr = initial_lr * lr_lambda(epoch); the
    constructor performs the initial step (epoch 0), every `step()` advances the epoch by one."""

    def __init__(self, optimizer, lr_lambda: Callable[[int], float], last_epoch: int = -1):
        self.optimizer, self.lr_lambda = optimizer, lr_lambda
        if last_epoch == -1:
            optimizer.initial_lr = optimizer.lr
        elif not hasattr(optimizer, "initial_lr"):
            raise KeyError("param 'initial_lr' is not specified in the optimizer when resuming a schedule")
        self.base_lr = float(optimizer.initial_lr)
        self.last_epoch = last_epoch
        self.step()

    def step(self) -> None:
        self.last_epoch += 1
        self._last_lr = self.base_lr * self.lr_lambda(self.last_epoch)
        self.optimizer.lr = self._last_lr

    def get_last_lr(self):
        return [self._last_lr]

    def state_dict(self) -> dict:
        return dict(last_epoch=self.last_epoch, base_lr=self.base_lr, _last_lr=self._last_lr)

    def load_state_dict(self, sd: dict) -> None:
        self.last_epoch, self.base_lr, self._last_lr = int(sd["last_epoch"]), float(sd["base_lr"]), float(sd["_last_lr"])
        self.optimizer.lr = self._last_lr


def _warm(step: int, num_warmup_steps: int) -> float:
    return float(step) / float(max(1, num_warmup_steps))


def get_constant_schedule(optimizer, last_epoch: int = -1) -> LambdaLR:
    return LambdaLR(optimizer, lambda _: 1, last_epoch=last_epoch)


def get_constant_schedule_with_warmup(optimizer, num_warmup_steps: int, last_epoch: int = -1) -> LambdaLR:
    # optimization.py:72-78: the warm-up divisor is max(1.0, warmup) here and max(1, warmup) in the others: same value
    return LambdaLR(optimizer, lambda s: _warm(s, num_warmup_steps) if s < num_warmup_steps else 1.0, last_epoch=last_epoch)


def get_piecewise_constant_schedule(optimizer, step_rules: str, last_epoch: int = -1) -> LambdaLR:
    """step_rules "1:10,0.1:20,0.005": multiplier 1 below step 10, 0.1 below step 20, 0.005 afterwards."""
    *rules, last = step_rules.split(",")
    table = sorted((int(st), float(v)) for v, st in (r.split(":") for r in rules))
    tail = float(last)

    def mult(step: int) -> float:
        for bound, value in table:
            if step < bound:
                return value
        return tail

    return LambdaLR(optimizer, mult, last_epoch=last_epoch)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, last_epoch: int = -1) -> LambdaLR:
    def mult(s: int) -> float:
        if s < num_warmup_steps:
            return _warm(s, num_warmup_steps)
        return max(0.0, float(num_training_steps - s) / float(max(1, num_training_steps - num_warmup_steps)))

    return LambdaLR(optimizer, mult, last_epoch)


def _progress(s: int, num_warmup_steps: int, num_training_steps: int) -> float:
    return float(s - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))


def get_cosine_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                                    last_epoch: int = -1) -> LambdaLR:
    def mult(s: int) -> float:
        if s < num_warmup_steps:
            return _warm(s, num_warmup_steps)
        p = _progress(s, num_warmup_steps, num_training_steps)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * p)))

    return LambdaLR(optimizer, mult, last_epoch)


def get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int,
                                                       num_cycles: int = 1, last_epoch: int = -1) -> LambdaLR:
    def mult(s: int) -> float:
        if s < num_warmup_steps:
            return _warm(s, num_warmup_steps)
        p = _progress(s, num_warmup_steps, num_training_steps)
        if p >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * p) % 1.0))))

    return LambdaLR(optimizer, mult, last_epoch)


def get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, lr_end: float = 1e-7,
                                              power: float = 1.0, last_epoch: int = -1) -> LambdaLR:
    lr_init = float(optimizer.lr)
    if not lr_init > lr_end:
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def mult(s: int) -> float:
        if s < num_warmup_steps:
            return _warm(s, num_warmup_steps)
        if s > num_training_steps:
            return lr_end / lr_init
        remaining = 1 - (s - num_warmup_steps) / (num_training_steps - num_warmup_steps)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init

    return LambdaLR(optimizer, mult, last_epoch)


def get_scheduler(name: Union[str, SchedulerType], optimizer, step_rules: Optional[str] = None,
                  num_warmup_steps: Optional[int] = None, num_training_steps: Optional[int] = None, num_cycles: int = 1,
                  power: float = 1.0, last_epoch: int = -1) -> LambdaLR:
    """optimization.py:288-362: one entry for every schedule, with the reference's argument checks."""
    name = SchedulerType(name)
    if name == SchedulerType.CONSTANT:
        return get_constant_schedule(optimizer, last_epoch=last_epoch)
    if name == SchedulerType.PIECEWISE_CONSTANT:
        return get_piecewise_constant_schedule(optimizer, step_rules=step_rules, last_epoch=last_epoch)
    if num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    if name == SchedulerType.CONSTANT_WITH_WARMUP:
        return get_constant_schedule_with_warmup(optimizer, num_warmup_steps=num_warmup_steps, last_epoch=last_epoch)
    if num_training_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    if name == SchedulerType.COSINE_WITH_RESTARTS:
        return get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, num_warmup_steps=num_warmup_steps,
                                                                  num_training_steps=num_training_steps, num_cycles=num_cycles,
                                                                  last_epoch=last_epoch)
    if name == SchedulerType.POLYNOMIAL:
        return get_polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps=num_warmup_steps,
                                                         num_training_steps=num_training_steps, power=power, last_epoch=last_epoch)
    fn = get_linear_schedule_with_warmup if name == SchedulerType.LINEAR else get_cosine_schedule_with_warmup
    return fn(optimizer, num_warmup_steps=num_warmup_steps, num_training_steps=num_training_steps, last_epoch=last_epoch)
