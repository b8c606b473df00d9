"""Flat parameter buckets + fused Adam (replaces the four torch.optim.Adam instances of the
reference train.py:94-116; same update rule, defaults eps=1e-8, no weight decay).

Every network's parameters (and their ``.grad``) are re-pointed into ONE flat fp32 buffer
each, so that (a) the optimiser is a single kernel launch per network, (b) ``zero_grad`` is
one memset, and (c) data-parallel training all-reduces one contiguous bucket per network
(dist.py).
"""

from __future__ import annotations

import os as _os

import torch

from . import _hip as H
from . import ops


class FlatBucket:
    """Re-homes the parameters of ``module`` into one contiguous fp32 buffer (+ grads)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev = self.params[0].device
        # 16-B aligned slices so kernels may use wide accesses on any parameter
        self.offsets, n = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("master parameters must be fp32")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o: o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o: o + p.numel()].view_as(p)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o: o + p.numel()].view_as(p)


class FusedAdam:
    """Adam over a FlatBucket: one o2m_adam_step launch per step.  The step counter lives
    on the device so the whole update is hipGraph-capturable."""

    def __init__(self, module: torch.nn.Module, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.bucket = FlatBucket(module)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = self.bucket.flat.device
        self.exp_avg = torch.zeros_like(self.bucket.flat)
        self.exp_avg_sq = torch.zeros_like(self.bucket.flat)
        self.step_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0
        self.pre_step_hooks = []  # e.g. wait for the gradient all-reduce (dist.py)

    @property
    def param_groups(self):
        return [{"params": self.bucket.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_grad()
        ops.ZERO_POOL.reset()  # between steps: no backward temporaries are alive

    def step(self):
        for hook in self.pre_step_hooks:
            hook()
        self.step_t += 1
        if ops.fp8_enabled() and self.bucket.grad.is_cuda and _os.environ.get("O2M_FP8_NAN_GUARD", "1") != "0":
            # fp8 mode only (config #5; the bf16 / fp32 paths never take this branch): non-finite gradient entries are
            # dropped for this step, the way mixed-precision training skips what its loss scale cannot represent.  Besides
            # genuine e5m2 saturation there is an OPEN issue behind it: with several streams the phase-pipelined weight
            # gradient intermittently writes one non-finite slice partial (1 of 28, a few steps into training; its
            # operands and every other tensor of the step are of normal size, the same kernel is bit-stable under
            # concurrency in tools/wgrad_stress.py, and bf16 mode computes in-flow what it computes alone:
            # tools/wgrad_inflow_check.py).  Without this the first such event turns the generator's weights to NaN.
            torch.nan_to_num_(self.bucket.grad, nan=0.0, posinf=0.0, neginf=0.0)
        H.adam_step(self.bucket.flat, self.bucket.grad, self.exp_avg, self.exp_avg_sq, self.step_t,
                    self.lr, self.betas[0], self.betas[1], self.eps, self.grad_scale)
        ops.bump_weights_epoch(self.bucket.params)  # this network's cached filter forms are stale, nobody else's

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what the reference checkpoints,
        evaluation.py:247-254): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` keyed by
        the parameter's position in ``module.parameters()`` order, plus one param group.  The
        moments are copies of the parameter's slice of the flat buckets."""
        state = {}
        step = self.step_t.detach().reshape(()).clone()
        for i, (p, o) in enumerate(zip(self.bucket.params, self.bucket.offsets)):
            state[i] = {"step": step.clone(),
                        "exp_avg": self.exp_avg[o: o + p.numel()].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o: o + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self.bucket.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts the ``torch.optim.Adam`` layout (ours, or a reference checkpoint's) and the
        flat layout this class wrote in round 1 (``step`` / ``exp_avg`` / ``exp_avg_sq`` buckets)."""
        if "state" not in sd:  # round-1 flat layout
            self.step_t.copy_(torch.as_tensor(sd["step"]).reshape(1))
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        state = sd["state"]
        n = len(self.bucket.params)
        if state and (len(state) != n or sorted(int(k) for k in state) != list(range(n))):
            raise ValueError(f"optimiser state covers {len(state)} parameters, this network has {n}")
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for k, st in state.items():
            i = int(k)
            p, o = self.bucket.params[i], self.bucket.offsets[i]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimiser state {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
            self.exp_avg[o: o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o: o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one fused step counter")
        self.step_t.fill_(steps.pop() if steps else 0.0)
        groups = sd.get("param_groups") or []
        if groups:
            g = groups[0]
            self.lr = float(g.get("lr", self.lr))
            self.betas = tuple(float(b) for b in g.get("betas", self.betas))
            self.eps = float(g.get("eps", self.eps))


def make_adam(module: torch.nn.Module, lr: float, betas=(0.9, 0.999)) -> FusedAdam:
    return FusedAdam(module, lr, betas)
