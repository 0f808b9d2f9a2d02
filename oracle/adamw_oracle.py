"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (clg_vqa_amd/*).

Plain-torch restatement of the reference's optimizer step (volta/train_task.py:316-343):

    loss/grad_acc -> backward -> clip_grad_norm_(model.parameters(), max_norm) -> AdamW.step -> scheduler.step -> zero_grad

with ``AdamW`` / ``WarmupLinearSchedule`` from ``pytorch_transformers.optimization`` (call sites train_task.py:264-276).
That package is an un-vendored dependency (``pytorch-transformers>=1.1.0`` in volta/requirements.txt), absent from the
reference tree and from the image, and the reference holds no test or golden vector for it:

    PARITY UNPINNED for the AdamW arithmetic -- restated from the package's published algorithm (v1.1/1.2
    ``optimization.py``: Adam moments, optional bias correction folded into the step size, decoupled weight decay
    applied AFTER the Adam update with the scheduled lr, eps added to sqrt(v) outside the bias correction).
    The schedule IS pinned: ``transformers.get_linear_schedule_with_warmup`` (importable here, the renamed successor of
    WarmupLinearSchedule) in tests/test_abi_and_host.py.

Parameter grouping restates train_task.py:249-260 (one group per tensor; lr 1e-4 for names with "vil_", weight decay 0
for names containing "bias" / "LayerNorm.bias" / "LayerNorm.weight").
"""
import math

import torch

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def warmup_linear(step, warmup_steps, t_total):
    """WarmupLinearSchedule.lr_lambda: step / max(1, warmup) while warming up, then linear decay to 0 at t_total."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


class ReferenceAdamW(object):
    def __init__(self, named_params, base_lr, weight_decay, betas=(0.9, 0.999), eps=1e-6, correct_bias=True,
                 warmup_steps=0, t_total=None, max_grad_norm=1.0):
        self.groups, seen = [], set()
        for name, p in named_params:
            if not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            self.groups.append(dict(name=name, p=p, lr=1e-4 if "vil_" in name else base_lr,
                                    wd=0.0 if any(nd in name for nd in NO_DECAY) else weight_decay,
                                    step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
        self.betas, self.eps, self.correct_bias = betas, eps, correct_bias
        self.warmup_steps, self.t_total, self.max_grad_norm = warmup_steps, t_total, max_grad_norm
        self.sched_step = 0  # LambdaLR.last_epoch: the lr in effect is base_lr * lambda(sched_step)

    def clip(self):
        """torch.nn.utils.clip_grad_norm_ (train_task.py:330): total L2 norm over all gradients; coefficient
        max_norm / (norm + 1e-6) clamped to 1."""
        grads = [g["p"].grad for g in self.groups if g["p"].grad is not None]
        total = torch.sqrt(sum((x.detach().double() ** 2).sum() for x in grads)).float()
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0)
        for x in grads:
            x.mul_(coef)
        return total

    @torch.no_grad()
    def step(self):
        norm = self.clip() if self.max_grad_norm is not None and math.isfinite(self.max_grad_norm) else None
        mult = 1.0 if self.t_total is None else warmup_linear(self.sched_step, self.warmup_steps, self.t_total)
        b1, b2 = self.betas
        for g in self.groups:
            p = g["p"]
            if p.grad is None:
                continue
            grad = p.grad
            g["step"] += 1
            g["m"].mul_(b1).add_(grad, alpha=1.0 - b1)
            g["v"].mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
            denom = g["v"].sqrt().add_(self.eps)
            lr = g["lr"] * mult
            step_size = lr
            if self.correct_bias:
                step_size = lr * math.sqrt(1.0 - b2 ** g["step"]) / (1.0 - b1 ** g["step"])
            p.addcdiv_(g["m"], denom, value=-step_size)
            if g["wd"] > 0.0:
                p.add_(p, alpha=-lr * g["wd"])
        self.sched_step += 1
        for g in self.groups:
            g["p"].grad = None
        return norm
