"""Step functions with the reference's signatures and return structure
(src/core/training.py).  Differences that do not change results:

* the generator forward of the discriminator step runs under ``no_grad`` (the reference
  builds that graph and throws it away, training.py:98-99);
* the discriminator's weight gradients are not computed in the generator step (the
  reference computes them and zeroes them at the next discriminator step, training.py:88);
* the logged scalars leave the device in ONE packed transfer per step function instead of
  3 + 7 blocking ``.item()`` calls (training.py:13,125-128,250-257);
* the discriminator scores the fake and the real batch in one 2B pass (per-sample network).
"""

from __future__ import annotations

import contextlib
import os
import random
from collections.abc import Iterator

import torch
import torch.nn.functional as F

from .. import ops
from ..model.loss import kl_loss_func, path_loss_func, path_loss_halves, style_cycle_loss_func


class ImageBuffer:
    """History pool of generated images (reference training.py:22-65).  Same Python
    ``random`` draw sequence as the reference: one ``uniform`` per image once the pool is
    full, plus one ``randint`` when the image is swapped."""

    def __init__(self, buffer_size: int):
        if buffer_size < 1:
            raise ValueError
        self.buffer_size = buffer_size
        self.num_imgs = 0
        self.images: list[torch.Tensor] = []

    def __call__(self, images: torch.Tensor):
        picked = []
        for k in range(images.shape[0]):
            fresh = images[k: k + 1].detach()
            if self.num_imgs < self.buffer_size:
                self.images.append(fresh)
                self.num_imgs += 1
                picked.append(fresh)
                continue
            if random.uniform(0, 1) > 0.5:
                slot = random.randint(0, self.buffer_size - 1)
                picked.append(self.images[slot].clone())
                self.images[slot] = fresh
            else:
                picked.append(fresh)
        return torch.cat(picked, 0)


class _PendingScalars:
    """The packed scalars of one step function on their way to the host: a non-blocking copy into pinned memory and
    an event behind it.  The first read waits for that event only."""

    __slots__ = ("host", "event", "values")

    def __init__(self, packed: torch.Tensor):
        self.host = torch.empty(packed.shape, dtype=torch.float32, pin_memory=True)
        self.host.copy_(packed, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(packed.device))
        self.values = None

    def get(self, i: int) -> float:
        if self.values is None:
            self.event.synchronize()
            self.values = self.host.tolist()
            self.host = None
        return self.values[i]


class LoggedScalar:
    """A logged loss / confidence whose device->host copy may still be in flight.  ``float(x)`` (and every arithmetic
    or comparison operator, ``format`` and ``math.*`` through it) waits for the copy on first use.  The reference reads
    ten scalars per step with blocking ``.item()`` calls (training.py:13,125-128,250-257) and appends them to the
    logger's lists; read this way the lists hold the same numbers at print time and the host is never made to wait
    for the step it has just queued (``set_async_scalars``)."""

    __slots__ = ("_pending", "_index")

    def __init__(self, pending: _PendingScalars, index: int):
        self._pending, self._index = pending, index

    def __float__(self):
        return self._pending.get(self._index)

    def __repr__(self):
        return repr(float(self))

    def __format__(self, spec):
        return format(float(self), spec)

    def __bool__(self):
        return bool(float(self))

    def __hash__(self):
        return hash(float(self))

    def __neg__(self):
        return -float(self)

    def __abs__(self):
        return abs(float(self))


def _scalar_op(name):
    def op(self, other):
        return getattr(float, name)(float(self), float(other))

    return op


for _n in ("add", "radd", "sub", "rsub", "mul", "rmul", "truediv", "rtruediv", "pow", "rpow", "lt", "le", "gt", "ge",
           "eq", "ne"):
    setattr(LoggedScalar, f"__{_n}__", _scalar_op(f"__{_n}__"))

_ASYNC_SCALARS = False


def set_async_scalars(flag: bool) -> None:
    """On: the step functions return ``LoggedScalar`` objects instead of floats (train.py and bench.py switch it on;
    off by default so that a caller written against the reference gets plain floats)."""
    global _ASYNC_SCALARS
    _ASYNC_SCALARS = bool(flag) and os.environ.get("O2M_ASYNC_SCALARS", "1") != "0"  # (=0: A/B runs)


_SCALAR_SINK = None  # core.graphed.GraphedStep: the logged scalars are summed on the device instead of returned


def _floats(*scalars):
    """One device->host transfer for all logged scalars of a step."""
    packed = torch.stack([s.detach().float().reshape(()) for s in scalars])
    if _SCALAR_SINK is not None:
        _SCALAR_SINK.add(packed)
        return [float("nan")] * len(scalars)  # (read them from the sink: ScalarSink.means_and_reset)
    if _ASYNC_SCALARS and packed.device.type == "cuda":
        pending = _PendingScalars(packed)
        return [LoggedScalar(pending, i) for i in range(len(scalars))]
    return packed.tolist()


def _l1(a, b):
    return ops.l1_sum(ops.to_internal(a), ops.to_internal(b)) / a.numel()


def _await_discriminator(device, discriminator):
    """The generator step's first use of the discriminator: behind the discriminator step's optimiser (which may still
    be running on its own stream), then its filter forms."""
    ops.d_step_wait(device, "done")
    ops.prepare_network(discriminator)


def _gan_loss(scores):
    """((D(G(x)) - 1)^2).mean() (training.py:202) through the fused patch-map reduction."""
    sums = ops.lsgan_sums(ops.to_internal(scores), scores.shape[0], 1.0, 1.0)
    return sums[0] / scores.numel()


@contextlib.contextmanager
def _frozen(module):
    flags = [p.requires_grad for p in module.parameters()]
    for p in module.parameters():
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in zip(module.parameters(), flags):
            p.requires_grad_(f)


def discriminator_step(config, device, discriminator, generator, mapping_network,
                       discriminator_optimiser, shoeprint_iter: Iterator[torch.Tensor],
                       shoemark_iter: Iterator[torch.Tensor], image_buffer, ada, ada_p):
    """One discriminator update; returns ``(loss, (real_confidence, fake_confidence))``
    exactly like the reference (training.py:71-128)."""
    st = ops.d_step_stream(device) if torch.device(device).type == "cuda" else None
    if st is None:
        return _discriminator_step(config, device, discriminator, generator, mapping_network, discriminator_optimiser,
                                   shoeprint_iter, shoemark_iter, image_buffer, ada, ada_p)
    # on its own stream, behind everything queued so far (the previous generator step's optimisers included): the
    # generator step that follows runs its generator-side passes beside this step's backward (ops.d_step_stream)
    st.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(st):
        out = _discriminator_step(config, device, discriminator, generator, mapping_network, discriminator_optimiser,
                                  shoeprint_iter, shoemark_iter, image_buffer, ada, ada_p)
        ops.d_step_mark(device, "done")
    return out


def _discriminator_step(config, device, discriminator, generator, mapping_network,
                        discriminator_optimiser, shoeprint_iter, shoemark_iter, image_buffer, ada, ada_p):
    batch = config["training"]["batch_size"]
    discriminator_optimiser.zero_grad()
    for net in (generator, discriminator):
        ops.prepare_network(net)  # every stale filter of a network in one launch
    ops.d_step_mark(device, "prep")  # (the generator's filter forms are shared with the generator step)

    shoeprints = next(shoeprint_iter).to(device)
    with torch.no_grad():
        w = mapping_network.get_single_w(batch, generator.n_style_blocks, device, 1)
        generated = generator(shoeprints, w)
    fake = ada(image_buffer(generated))
    real = ada(next(shoemark_iter).to(device))

    # one 2B pass instead of the reference's two B passes (training.py:107-108): every op of
    # the discriminator is per-sample (InstanceNorm, no batch statistics), so the scores and
    # the gradients are the same, with half the launches and twice the rows per GEMM
    both = discriminator(torch.cat([fake.float(), real.float()], dim=0))
    # LSGAN loss of both halves and the two confidences from ONE launch over the 2B patch maps (o2m_lsgan_fwd)
    sums = ops.lsgan_sums(ops.to_internal(both), batch, 0.0, 1.0)
    n = both[:batch].numel()
    loss = ops.weighted_sum((sums[0], sums[1]), (0.5 / n, 0.5 / n))
    conf = sums.detach()
    sign_fake, sign_real = conf[2] * (-1.0 / n), conf[3] * (1.0 / n)
    ada_p.update_p(sign_real)

    loss.backward()
    discriminator_optimiser.step()
    # the updated discriminator's filter forms here, on this step's stream: the generator step then finds them fresh when
    # it reaches its adversarial term (_await_discriminator) instead of building them on its own critical path
    ops.prepare_network(discriminator)
    out = _floats(loss, sign_real, sign_fake)
    return out[0], (out[1], out[2])



# O2M_PATH_TAP=0: path-loss terms from the collected feature maps after the pass (two consumers per map: autograd adds)
_PATH_TAP = os.environ.get("O2M_PATH_TAP", "1") == "1"
_BATCH_DECODES = os.environ.get("O2M_BATCH_DECODES", "1") == "1"
# O2M_SIDE_STYLE=0: both style-extractor passes of generator_step on the main stream (A/B of running them beside the
# encoder / the discriminator on the second stream)
_SIDE_STYLE = os.environ.get("O2M_SIDE_STYLE", "1") == "1"
# O2M_SIDE_MAPPING=1 (experiment, off): generator_step's mapping-network passes on a stream of their own.  Measured
# SLOWER: a chain of ~60 dependent 4-us launches beside kernels that keep every CU busy advances one link per freed CU --
# 35.6-35.8 vs 34.4 ms per step, 38.3 with eight hardware queues (GPU_MAX_HW_QUEUES=8), 42.4 on a high-priority stream
# (O2M_PRIO_MAP=-1), the decode group waiting for it all the while (profiles/r04_ab_mapping_stream.txt)
_SIDE_MAPPING = os.environ.get("O2M_SIDE_MAPPING", "0") == "1"


def _separate_decodes(config, device, generator, discriminator, mapping_network, style_extractor, ada, latents,
                      shoeprints, shoemarks):
    """The reference's call sequence (training.py:171-234), one decoder pass per term."""
    batch, lam, blocks = config["training"]["batch_size"], config["optimisation"], generator.n_style_blocks
    z_print, z_mark = latents.chunk(2, dim=0)
    w_zero = mapping_network.get_single_w(batch, blocks, device, 0)
    rec = _l1(generator.decode(z_print, w_zero), shoeprints)
    w_mark = style_extractor(shoemarks)
    idt = _l1(generator.decode(z_mark, w_mark.expand(blocks, *w_mark.shape)), shoemarks)
    w_trans = mapping_network.get_single_w(batch, blocks, device, 1)
    generated = generator.decode(z_print, w_trans)
    _await_discriminator(device, discriminator)
    with _frozen(discriminator):
        gan = _gan_loss(discriminator(ada(generated)))
    style = style_cycle_loss_func(w_trans[-1], style_extractor(generated))
    # (device draw when the mapping network draws there too: core/graphed.py)
    theta = torch.rand(batch, device=device) if getattr(mapping_network, "device_draws", False) else ops.host_to_device(torch.rand(batch), device)
    lo, hi = lam["path_loss_jacobian_granularity"]
    h = torch.ones_like(theta).uniform_(lo, hi)
    d1, d2 = (theta + h / 2).clamp(0, 1), (theta - h / 2).clamp(0, 1)
    w1, w2 = mapping_network.get_two_w(batch, blocks, device, (d1, d2))
    path = path_loss_func(generator.extract(z_print, w1), generator.extract(z_print, w2), h)
    return rec, idt, gan, style, path


def _batched_decodes(config, device, generator, discriminator, mapping_network, style_extractor, ada, t_lat,
                     shoeprints, shoemarks, w_mark=None, mark_ready=None, step_start=None):
    """The same five decoder passes as TWO: the three decodes (training.py:171-199) as one 3B batch and the two
    feature extractions (training.py:226-231) as one 2B batch.  The decoder is per-sample throughout (style
    modulation / demodulation per sample, no normalisation), so every sample's result is that of its separate
    call -- with 2 instead of 5 launches per layer and each filter's weight gradient reduced once per group.
    ``t_lat``: internal [2B, h, w, C] latents, chunk 0 = shoeprints, chunk 1 = shoemarks."""
    batch, lam, blocks = config["training"]["batch_size"], config["optimisation"], generator.n_style_blocks
    # Every style vector first, in the reference's draw order (builder.py:115-128, training.py:214-223:
    # get_single_w(1) -> theta -> h -> get_two_w; get_single_w(0) and the style extractor draw nothing).  With the
    # augmentation ON (p > 0) ``ada(generated)`` draws from the CPU generator too, and here it runs AFTER theta and
    # get_two_w, while the reference (and _separate_decodes) has it between get_single_w(1) and theta: the batched form
    # reorders the CPU draws then (statistically harmless; the benchmark and the parity cases hold p at 0).
    side = ops.group_stream(device)

    def draw_styles():
        w_zero = mapping_network.get_single_w(batch, blocks, device, 0)
        w_trans = mapping_network.get_single_w(batch, blocks, device, 1)
        # (device draw when the mapping network draws there too: core/graphed.py)
        theta = (torch.rand(batch, device=device) if getattr(mapping_network, "device_draws", False)
                 else ops.host_to_device(torch.rand(batch), device))
        lo, hi = lam["path_loss_jacobian_granularity"]
        h = torch.ones_like(theta).uniform_(lo, hi)
        d1, d2 = (theta + h / 2).clamp(0, 1), (theta - h / 2).clamp(0, 1)
        w1, w2 = mapping_network.get_two_w(batch, blocks, device, (d1, d2))
        return w_zero, w_trans, h, w1, w2

    mst = ops.mapping_stream(device) if _SIDE_MAPPING else None
    if mst is not None:
        # the mapping network's ~60 tiny launches (and, in backward, its ~100) on a stream of their own: beside the
        # encoder's forward, and in backward beside the encoder's backward instead of in front of it on the main stream
        # (the host draws, and therefore the random numbers, are in the same order).  NOT the group stream: there they
        # queue behind the style extractor's pass and the decode group waits for them (+1.6 ms)
        main0 = torch.cuda.current_stream(device)
        if step_start is not None:
            mst.wait_event(step_start)  # (behind the previous step's optimisers, NOT behind the encoder queued since)
        else:
            mst.wait_stream(main0)
        with torch.cuda.stream(mst):
            w_zero, w_trans, h, w1, w2 = draw_styles()
            styles_ready = torch.cuda.Event()
            styles_ready.record(mst)
        main0.wait_event(styles_ready)
        for t in (w_zero, w_trans, h, w1, w2):
            t.record_stream(main0)
            if side is not None:
                t.record_stream(side)
    else:
        w_zero, w_trans, h, w1, w2 = draw_styles()
    if w_mark is None:
        w_mark = style_extractor(shoemarks)
    else:  # computed on the second stream beside the encoder (generator_step): order the main stream behind it
        torch.cuda.current_stream(device).wait_event(mark_ready)
        w_mark.record_stream(torch.cuda.current_stream(device))

    # The extraction group depends on the latents and the style vectors only: it runs on a second stream, so its
    # HBM-bound kernels share the chip with the decode group's MFMA kernels (and the reverse) in forward AND in
    # backward (autograd runs a node's backward on the stream of its forward and orders the streams itself).

    def extraction_group():
        w_ext = torch.cat([w1, w2], dim=1)
        # the path-loss term of every feature map is taken as the map passes (ops.halves_sq_tap)
        inv_h2 = (1.0 / (h.float() ** 2)).contiguous()
        feats = generator._decode(ops.batch_gather(t_lat, batch, (0, 0)), w_ext, collect=True, internal=True,
                                  tap=inv_h2 if _PATH_TAP else None)
        return path_loss_halves(feats, h)

    if side is not None:
        main = torch.cuda.current_stream(device)
        generator.prepare_decoder_weights()  # both groups read the cached filter forms: build them before the fork
        side.wait_stream(main)
        with torch.cuda.stream(side):
            path = extraction_group()

    w_dec = torch.cat([w_zero, w_mark.expand(blocks, *w_mark.shape), w_trans], dim=1)
    images = generator._decode(ops.batch_gather(t_lat, batch, (0, 1, 0)), w_dec, collect=False, internal=True)
    rec_t, idt_t, gen_t = ops.split_batch(images, 3)
    n_img = shoeprints.numel()
    rec = ops.l1_sum(rec_t, ops.to_internal(shoeprints)) / n_img
    idt = ops.l1_sum(idt_t, ops.to_internal(shoemarks)) / n_img
    generated = ops.to_public(gen_t, shoeprints.shape[1])
    style_of_generated = style = None
    if side is not None and _SIDE_STYLE:
        # the style extractor's pass over the generated images beside the discriminator's (independent networks,
        # one input): on the second stream, behind the extraction group
        made = torch.cuda.Event()
        made.record(main)
        side.wait_event(made)
        gen_t.record_stream(side)
        with torch.cuda.stream(side):
            style_of_generated = style_extractor(generated)
            # its loss term there too: the ~15 tiny kernels of the cosine / MSE arithmetic (and, in backward, their ~40)
            # run beside the discriminator instead of in the stretch between the forward and backward passes where
            # nothing else can
            style = style_cycle_loss_func(w_trans[-1], style_of_generated)
    _await_discriminator(device, discriminator)
    with _frozen(discriminator):
        gan = _gan_loss(discriminator(ada(generated)))

    if style_of_generated is None:
        style_of_generated = style_extractor(generated)
    if side is None:
        path = extraction_group()
    else:
        main.wait_stream(side)
        path.record_stream(main)
        style_of_generated.record_stream(main)
        if style is not None:
            style.record_stream(main)
    if style is None:
        style = style_cycle_loss_func(w_trans[-1], style_of_generated)
    return rec, idt, gan, style, path


def generator_step(config, device, generator, discriminator, mapping_network, style_extractor,
                   generator_optimiser, mapping_network_optimiser, style_extractor_optimiser,
                   shoeprint_iter: Iterator[torch.Tensor], shoemark_iter: Iterator[torch.Tensor], ada,
                   *, kl_moment_hook=None):
    """One generator / mapping-network / style-extractor update; returns
    ``(total, (gan, rec, idt, kl, path, style))`` like the reference (training.py:136-257)."""
    batch = config["training"]["batch_size"]
    lam = config["optimisation"]
    blocks = generator.n_style_blocks
    # a discriminator step running on its own stream (ops.d_step_stream) has rebuilt the generator's filter forms: wait
    # for those; the discriminator itself (its optimiser still running there) is waited for and prepared where the
    # adversarial term needs it (_await_discriminator)
    ops.d_step_wait(device, "prep")
    for opt in (generator_optimiser, mapping_network_optimiser, style_extractor_optimiser):
        opt.zero_grad()
    for net in (generator, style_extractor):
        ops.prepare_network(net)  # every stale filter of a network in one launch

    shoeprints = next(shoeprint_iter).to(device)
    shoemarks = next(shoemark_iter).to(device)

    # the style of the real shoemarks does not depend on the generator: its pass runs on the second stream beside
    # the encoder (and its backward beside whatever the main stream does then)
    w_mark = mark_ready = step_start = None
    if torch.device(device).type == "cuda" and not torch.cuda.is_current_stream_capturing():
        step_start = torch.cuda.Event()
        step_start.record(torch.cuda.current_stream(device))
    side = ops.group_stream(device) if (_BATCH_DECODES and _SIDE_STYLE) else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            w_mark = style_extractor(shoemarks)
            mark_ready = torch.cuda.Event()
            mark_ready.record(side)

    # both domains through the encoder in one 2B pass; KL on the joint latent
    latents = generator.encode(torch.cat([shoeprints, shoemarks], dim=0))
    kl = kl_loss_func(latents, moment_hook=kl_moment_hook)
    if config["architecture"]["add_latent_noise"]:
        latents = latents + torch.randn_like(latents)
    t_lat = ops.to_internal(latents)  # [2B, h, w, C]: chunk 0 = shoeprint latents, chunk 1 = shoemark latents

    if not _BATCH_DECODES:  # O2M_BATCH_DECODES=0: the five decoder passes one by one (A/B of the batched form)
        rec, idt, gan, style, path = _separate_decodes(config, device, generator, discriminator, mapping_network,
                                                       style_extractor, ada, latents, shoeprints, shoemarks)
    else:
        rec, idt, gan, style, path = _batched_decodes(config, device, generator, discriminator, mapping_network,
                                                      style_extractor, ada, t_lat, shoeprints, shoemarks, w_mark, mark_ready,
                                                      step_start)

    total = ops.weighted_sum((gan, idt, rec, kl, path, style),
                             (1.0, lam["identity_loss_lambda"], lam["reconstruction_loss_lambda"], lam["kl_loss_lambda"],
                              lam["path_loss_lambda"], lam["style_cycle_loss_lambda"]))
    total.backward()
    generator_optimiser.step()
    mapping_network_optimiser.step()
    style_extractor_optimiser.step()
    out = _floats(total, gan, rec, idt, kl, path, style)
    return out[0], tuple(out[1:])
