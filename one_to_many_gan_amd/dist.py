"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL
(torch.distributed backend "nccl") over xGMI.  New relative to the reference, which is
single-device.

The step shards by SAMPLES: InstanceNorm and style modulation are per-sample, so each rank
runs the full D+G step on its own B/N images and only gradients are exchanged:

* one flat fp32 bucket per network (optim.FlatBucket) -> ONE all-reduce per network per
  step: D 11 MB in the D step; G 36 MB, S 11 MB, M 336 B in the G step;
* the all-reduce is launched from autograd (post-accumulate-grad hooks) the moment the last
  gradient of a bucket has been accumulated, on a side HIP stream ordered behind the
  backward stream by an event, so it overlaps the rest of backward; ``optimiser.step``
  waits on the completion event, and the 1/N averaging is folded into the Adam kernel;
* kl_loss_func uses the moments of the WHOLE batch (loss.py:86-87), which is not a mean of
  per-rank losses: ``kl_moment_hook`` all-reduces the two sums (2 floats) so every rank
  evaluates the global-batch KL, and scales its gradient by N to survive the averaging.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class BucketReducer:
    """Overlapped all-reduce of one FusedAdam's flat gradient bucket."""

    def __init__(self, optimiser, group=None):
        self.opt = optimiser
        self.bucket = optimiser.bucket
        self.group = group
        self.world = dist.get_world_size(group)
        self.pending = len(self.bucket.params)
        self.left = self.pending
        self.launched = False
        self.seen = set()
        self.work = None
        dev = self.bucket.flat.device
        self.on_gpu = dev.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
        self.done = torch.cuda.Event() if self.on_gpu else None
        self.enabled = True
        optimiser.grad_scale = 1.0 / self.world
        optimiser.pre_step_hooks.append(self.wait)
        from . import ops

        for p in self.bucket.params:
            if p.dim() == 4:
                # conv filters: their gradient is written by ops._finalize_weight_grads at the end
                # of backward (kernel-layout accumulators), not by autograd's AccumulateGrad --
                # whose post-accumulate hook can still fire for them, earlier, and must not count
                ops.GRAD_READY_HOOKS[p] = self._on_grad
            else:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _on_grad(self, param):
        if not self.enabled or id(param) in self.seen:
            return
        self.seen.add(id(param))
        self.left -= 1
        if self.left == 0:
            self._launch()

    def _launch(self):
        if self.launched:
            return
        self.launched = True
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.bucket.grad.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                dist.all_reduce(self.bucket.grad, op=dist.ReduceOp.SUM, group=self.group)
                self.done.record(self.comm_stream)
        else:  # gloo / CPU: used by the world_size-2 tests
            self.work = dist.all_reduce(self.bucket.grad, op=dist.ReduceOp.SUM, group=self.group,
                                        async_op=True)

    def wait(self):
        """Called by FusedAdam.step(): make sure the reduced gradients are visible."""
        if not self.enabled:
            return
        if not self.launched:  # some parameter got no gradient this step: reduce now
            self._launch()
        if self.on_gpu:
            torch.cuda.current_stream(self.bucket.grad.device).wait_event(self.done)
        elif self.work is not None:
            self.work.wait()
            self.work = None
        self.left = self.pending
        self.launched = False
        self.seen.clear()


def broadcast_parameters(optimisers, src: int = 0, group=None):
    """Identical starting weights on every rank (one broadcast per flat bucket)."""
    for opt in optimisers:
        dist.broadcast(opt.bucket.flat, src=src, group=group)


def make_kl_moment_hook(group=None):
    """moment_hook for model.loss.kl_loss_func: global-batch moments under data parallelism."""
    world = dist.get_world_size(group)

    class _AllReduceSum(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            out = t.clone()
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
            return out

        @staticmethod
        def backward(ctx, g):
            # every rank holds the same loss; gradients are averaged over ranks afterwards,
            # so each rank's share of d loss / d (its local sums) is scaled back up by N
            return g * world

    def hook(s1, s2, n):
        both = _AllReduceSum.apply(torch.stack([s1, s2]))
        return both[0], both[1], n * world

    return hook
