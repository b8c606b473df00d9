"""The D+G step as ONE HIP graph (SURVEY.md section 8(e) "scaling risk": Python dispatch, ``.item()`` syncs and
CPU-RNG sampling off the critical path).

The reference's loop body (train.py:204-251) depends on the host in five places; a step that is to be captured
once and replayed has to take each of them to the device:

* the style draws of ``MappingNetwork`` (builder.py:52-132: mix decision, crossover, z on the CPU generator) ->
  ``MappingNetwork.device_draws``: the same distributions drawn by the device generator, the crossover applied as a
  mask instead of a Python branch;
* ``theta`` of ``generator_step`` (training.py:214) -> drawn on the device when the mapping network draws there;
* ``ImageBuffer`` (training.py:22-65: Python ``random``) -> ``DeviceImageBuffer``: the pool is one tensor, the
  per-image decisions are device scalars, the images of a batch are still visited in order (a later image of the
  batch can draw a slot an earlier one has just filled, as in the reference);
* ``ADAp`` (loss.py:11-52: a Python list of scores, ``.item()``) -> ``DeviceADAp``: the same controller, window quirk
  included, as device arithmetic; the augmentation itself must be the identity (p held at 0, as in the benchmark);
* the ten logged scalars (training.py:125-128,250-257) -> ``ScalarSink``: summed on the device, read when a line is
  printed (the ``Logger`` prints window means, evaluation.py:45-53).

``GraphedStep`` runs the unchanged ``discriminator_step`` / ``generator_step`` eagerly until the history pool is
full, captures ONE step (both functions, all three streams, backward, the four Adam updates) with
``torch.cuda.graph`` and replays it; the input batches are copied into static buffers before every replay.

Measured (tools/graph_probe.py): 64x64x1, batch 4 -- the host-bound shape -- 14.5 -> 7.3 ms
per step; 256x256x3, batch 16: 41.6 -> 43.9 ms (the replayed three-stream schedule packs worse than the eager one),
so the graph is for the small shapes and the default loop stays eager.
"""

from __future__ import annotations

import torch

from . import training as _training


class DeviceImageBuffer:
    """History pool of generated images with the reference's per-image rule (training.py:39-65) on device state:
    while the pool fills, images are stored and returned (host-side count, exactly as the reference); once full,
    each image of the batch IN ORDER is swapped with a random slot with probability 1/2."""

    def __init__(self, buffer_size: int):
        if buffer_size < 1:
            raise ValueError
        self.buffer_size = buffer_size
        self.num_imgs = 0
        self.pool = None

    @property
    def full(self) -> bool:
        return self.num_imgs >= self.buffer_size

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        images = images.detach()
        if self.pool is None:
            self.pool = torch.zeros((self.buffer_size, *images.shape[1:]), dtype=images.dtype, device=images.device)
        out = []
        for k in range(images.shape[0]):
            fresh = images[k: k + 1]
            if self.num_imgs < self.buffer_size:
                self.pool[self.num_imgs: self.num_imgs + 1].copy_(fresh)
                self.num_imgs += 1
                out.append(fresh)
                continue
            swap = torch.rand((), device=images.device) > 0.5
            slot = torch.randint(0, self.buffer_size, (1,), device=images.device)
            old = self.pool.index_select(0, slot)
            out.append(torch.where(swap, old, fresh))
            self.pool.index_copy_(0, slot, torch.where(swap, fresh, old))
        return torch.cat(out, 0)

    # ---- the reference object, for checkpoints (evaluation.model_checkpoint / load_checkpoint)
    def to_reference(self) -> "_training.ImageBuffer":
        ref = _training.ImageBuffer(self.buffer_size)
        if self.pool is not None:
            ref.images = [self.pool[k: k + 1].clone() for k in range(self.num_imgs)]
        ref.num_imgs = self.num_imgs
        return ref

    def load_reference(self, ref, device):
        self.num_imgs = min(len(ref.images), self.buffer_size)
        if self.num_imgs:
            first = ref.images[0]
            self.pool = torch.zeros((self.buffer_size, *first.shape[1:]), dtype=first.dtype, device=device)
            for k in range(self.num_imgs):
                self.pool[k: k + 1].copy_(ref.images[k])


class DeviceADAp:
    """``ADAp`` (loss.py:11-52) as device arithmetic: same window, same step, same quirk (the score that closes a
    window also opens the next one), no upper clamp.  ``__call__`` returns 0.0 without reading the device -- the
    graph mode holds the augmentation at the identity; ``value()`` reads p."""

    def __init__(self, ada_e: float, ada_adjustment_size: float, batch_size: int,
                 discriminator_overfitting_target: float, device):
        self.window, self.step, self.target = ada_e // batch_size, ada_adjustment_size * ada_e, \
            discriminator_overfitting_target
        z = lambda: torch.zeros((), dtype=torch.float32, device=device)  # noqa: E731
        self.p, self.s_sum, self.n, self.curr = z(), z(), z(), z()

    def update_p(self, mean_score: torch.Tensor):
        score = mean_score.detach().float().reshape(())
        self.s_sum += score
        self.n += 1
        closing = self.curr == float(self.window)
        verdict = self.s_sum / self.n
        delta = torch.where(verdict > self.target, self.step, torch.where(verdict < self.target, -self.step, 0.0))
        self.p.copy_(torch.where(closing, torch.clamp_min(self.p + delta, 0.0), self.p))
        self.s_sum.copy_(torch.where(closing, score, self.s_sum))
        self.n.copy_(torch.where(closing, torch.ones_like(self.n), self.n))
        self.curr.copy_(torch.where(closing, torch.zeros_like(self.curr), self.curr))
        self.curr += 1

    def __call__(self) -> float:
        return 0.0

    def value(self) -> float:
        return float(self.p)

    # ---- the reference object, for checkpoints: same p, same position in the window, same window mean and count
    def to_reference(self):
        from ..model.loss import ADAp

        ref = ADAp(1, 0.0, 1, self.target)
        ref.window, ref.step = self.window, self.step
        ref.p = self.p.detach().cpu().clone()
        ref.curr_batch = int(self.curr)
        n = int(self.n)
        ref.mean_real_scores = [torch.tensor(float(self.s_sum) / n)] * n if n else []
        return ref

    def load_reference(self, ref):
        self.p.fill_(float(ref.p))
        self.curr.fill_(float(ref.curr_batch))
        self.n.fill_(float(len(ref.mean_real_scores)))
        self.s_sum.fill_(float(sum(float(s) for s in ref.mean_real_scores)))


class ScalarSink:
    """Device-side sums of the logged scalars of the two step functions (3 + 7 per step) and a step count."""

    def __init__(self, device):
        self.sums = {3: torch.zeros(3, dtype=torch.float32, device=device),
                     7: torch.zeros(7, dtype=torch.float32, device=device)}
        self.count = torch.zeros((), dtype=torch.float32, device=device)

    def add(self, packed: torch.Tensor):
        self.sums[packed.numel()] += packed
        if packed.numel() == 7:  # the generator step closes a D+G step
            self.count += 1

    def means_and_reset(self):
        """((d_loss, real_acc, fake_acc), (total, gan, rec, idt, kl, path, style)) averaged over the steps since
        the last call -- one blocking read."""
        n = max(float(self.count), 1.0)
        d, g = (self.sums[3] / n).tolist(), (self.sums[7] / n).tolist()
        for t in (*self.sums.values(), self.count):
            t.zero_()
        return tuple(d), tuple(g)


class GraphedStep:
    """``step()``: one D+G step (eager until the history pool is full, then a replay of the captured graph).

    ``shoeprint_iter`` / ``shoemark_iter`` yield device batches; each step draws two of each (one per step function),
    copied into static buffers that the captured step reads."""

    def __init__(self, config, device, nets, opts, shoeprint_iter, shoemark_iter, ada, *, warmup_steps: int = 3,
                 capture: bool = True):
        if device.type != "cuda":
            raise RuntimeError("GraphedStep needs a GPU")
        if getattr(ada, "p", 0.0) != 0.0 or type(ada).__name__ != "IdentityADA":
            raise NotImplementedError("the graphed step holds the augmentation at the identity (p = 0)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError("the graphed step is single-process (the bucket reducer launches from host hooks)")
        self.config, self.device, self.nets, self.opts, self.ada = config, device, nets, opts, ada
        t, a = config["training"], config["ada"]
        self.buffer = DeviceImageBuffer(t["image_buffer_size"])
        self.ada_p = DeviceADAp(a["ada_overfitting_measurement_n_images"], a["ada_adjustment_size"], t["batch_size"],
                                a["discriminator_real_acc_target"], device)
        self.sink = ScalarSink(device)
        self.src = (shoeprint_iter, shoemark_iter)
        self.static = None
        self.graph = None
        self.warmup_left = warmup_steps
        self.capture = capture  # False: the same device-resident step, never captured (the tests' reference run)
        nets["M"].device_draws = True

    def _fill(self):
        fresh = [next(self.src[0]), next(self.src[1]), next(self.src[0]), next(self.src[1])]  # D: print, mark; G: print, mark
        if self.static is None:
            self.static = [torch.empty_like(f, device=self.device) for f in fresh]
        for s, f in zip(self.static, fresh):
            s.copy_(f, non_blocking=True)

    def _body(self):
        n, o, c, dev = self.nets, self.opts, self.config, self.device
        prev = _training._SCALAR_SINK
        _training._SCALAR_SINK = self.sink
        try:
            _training.discriminator_step(c, dev, n["D"], n["G"], n["M"], o["D"], iter((self.static[0],)),
                                         iter((self.static[1],)), self.buffer, self.ada, self.ada_p)
            _training.generator_step(c, dev, n["G"], n["D"], n["M"], n["S"], o["G"], o["M"], o["S"],
                                     iter((self.static[2],)), iter((self.static[3],)), self.ada)
        finally:
            _training._SCALAR_SINK = prev

    def step(self):
        self._fill()
        if self.graph is not None:
            self.graph.replay()
            return
        if not self.capture or not self.buffer.full or self.warmup_left > 0:
            self._body()
            if self.buffer.full:
                self.warmup_left -= 1
            return
        # capture: on a side stream, as torch.cuda.graph requires; the step's own streams fork from and join it
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._body()
        self.graph = g
        # the capture pass does not execute: this step's work is the first replay
        self.graph.replay()

    def logged_means(self):
        return self.sink.means_and_reset()
