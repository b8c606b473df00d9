"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL
(torch.distributed backend "nccl") over xGMI.  New relative to the reference, which is
single-device.

The step shards by SAMPLES: InstanceNorm and style modulation are per-sample, so each rank
runs the full D+G step on its own B/N images and only gradients are exchanged:

* one flat fp32 bucket per network (optim.FlatBucket), all-reduced in SEGMENTS of whole
  parameters (default ~8 MB: D 2, G 5, S 2, M 1 segments at 256x256).  A segment's all-reduce is
  launched from inside backward the moment its last gradient is complete -- conv filters report
  that when their last use of the pass has been reduced (ops._finalize_layer), everything else
  through autograd's post-accumulate-grad hooks -- on a side HIP stream ordered behind the
  backward stream by an event.  Backward runs decoder -> encoder, parameters are registered
  encoder -> decoder, so the last segments go first and only the first one (the encoder stem) is
  exposed after backward.  Segments launch in one fixed order on every rank (last to first), as
  collectives require.  ``optimiser.step`` waits on the completion event; the 1/N averaging is
  folded into the Adam kernel;
* kl_loss_func uses the moments of the WHOLE batch (loss.py:86-87), which is not a mean of
  per-rank losses: ``kl_moment_hook`` all-reduces the two sums (2 floats) so every rank
  evaluates the global-batch KL, and scales its gradient by N to survive the averaging;
* ``ADAp`` must see the same discriminator confidence on every rank or the augmentation
  probability drifts apart (training.py:116-120, loss.py:32-49): ``sync_ada_p`` averages the
  score over ranks (1 float) before the controller consumes it.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class BucketReducer:
    """Overlapped, segmented all-reduce of one FusedAdam's flat gradient bucket."""

    def __init__(self, optimiser, group=None, segment_bytes: int = 8 << 20):
        self.opt = optimiser
        self.bucket = optimiser.bucket
        self.group = group
        self.world = dist.get_world_size(group)
        dev = self.bucket.flat.device
        self.on_gpu = dev.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
        self.done = torch.cuda.Event() if self.on_gpu else None
        self.enabled = True
        # segments: runs of whole parameters, in registration order, of >= segment_bytes each
        params, offs = self.bucket.params, self.bucket.offsets
        self.seg_of, self.seg_range, start, first = {}, [], 0, 0
        for i, p in enumerate(params):
            end = offs[i + 1] if i + 1 < len(params) else self.bucket.numel
            self.seg_of[id(p)] = len(self.seg_range)
            if (end - start) * 4 >= segment_bytes or i + 1 == len(params):
                self.seg_range.append((start, end, i + 1 - first))
                start, first = end, i + 1
        self.n_seg = len(self.seg_range)
        self.launch_log = []  # (segment, "hook" | "wait") of the CURRENT step (cleared by _reset), for the tests
        self.last_launch_log = []  # the finished step's log
        self._reset()
        optimiser.grad_scale = 1.0 / self.world
        optimiser.pre_step_hooks.append(self.wait)
        from . import ops

        for p in params:
            # conv filters: their gradient is written by ops._finalize_layer (kernel-layout
            # accumulators), not by autograd's AccumulateGrad -- whose post-accumulate hook can
            # still fire for them, earlier, and must not count.  to_style parameters may take either
            # route (style_bwd adds into .grad directly when it can): both report here, once.
            ops.GRAD_READY_HOOKS[p] = self._on_grad
            if p.dim() != 4:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _reset(self):
        # the log is per step: a 150k-step run would otherwise keep ~10 tuples per step per rank forever
        self.last_launch_log, self.launch_log = self.launch_log, []
        self.left = [n for _, _, n in self.seg_range]
        self.next = self.n_seg - 1  # segments go out last -> first
        self.seen = set()
        self.works = []

    def _on_grad(self, param):
        if not self.enabled or id(param) in self.seen:
            return
        self.seen.add(id(param))
        self.left[self.seg_of[id(param)]] -= 1
        self._launch_ready("hook")

    def _launch_ready(self, why, force=False):
        while self.next >= 0 and (force or self.left[self.next] == 0):
            self._launch(self.next, why)
            self.next -= 1

    def _launch(self, seg, why):
        a, b, _ = self.seg_range[seg]
        view = self.bucket.grad[a:b]
        self.launch_log.append((seg, why))
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(view.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                if seg == 0:
                    self.done.record(self.comm_stream)
        else:  # gloo / CPU: used by the world_size-2 tests
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Called by FusedAdam.step(): make sure the reduced gradients are visible."""
        if not self.enabled:
            return
        self._launch_ready("wait", force=True)  # parameters that got no gradient this step
        if self.on_gpu:
            torch.cuda.current_stream(self.bucket.grad.device).wait_event(self.done)
        for w in self.works:
            w.wait()
        self._reset()


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


def sync_ada_p(ada_p, group=None):
    """Make ``ada_p.update_p`` consume the discriminator confidence of the GLOBAL batch: the mean
    over ranks of each rank's ``sign(D(real))`` mean (equal local batches), so every rank's
    controller takes the same decisions and ``p`` stays identical across ranks."""
    world = dist.get_world_size(group)
    local_update = ada_p.update_p

    def update_p(mean_score: torch.Tensor):
        score = mean_score.detach().float().clone()
        dist.all_reduce(score, op=dist.ReduceOp.SUM, group=group)
        local_update(score / world)

    ada_p.update_p = update_p
    ada_p.local_update_p = local_update  # the unsynchronised controller (a rank working alone: bench.py's timer steps)
    return ada_p
