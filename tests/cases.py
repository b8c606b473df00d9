"""Parity cases shared by three consumers:

* ``tools/make_golden.py``  -- runs them on the *reference* (build container only) and
  writes ``tests/golden/*.npz``;
* ``tests/test_oracle_golden.py`` -- runs them on the CPU oracle and checks the fixtures;
* ``tests/test_hip_parity.py`` -- runs them on the HIP path (``-m gpu``) and checks both
  the fixtures and the live oracle.

A case is ``fn(ns, device) -> dict[str, Tensor]`` where ``ns`` is an implementation
namespace (``namespaces.py``) giving uniform constructors over the three code bases.
All inputs and weights are closed-form (oracle/detweights.py), so fixtures hold outputs only.
"""

from __future__ import annotations

import contextlib
import random
from collections import OrderedDict

import torch

from oracle import detweights as _dw
from oracle.detweights import fill_state_dict

_FP64 = {"on": False}


def host_threads() -> int:
    """Threads for the CPU oracle: the cores this process may run on, at most 16 (a GPU box reports
    every core of the host through os.cpu_count() but gives one GPU a 16-core share -- asking torch
    for 256 threads there made the oracle several times slower)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def _maybe_double(fn):
    def wrapped(*a, **k):
        t = fn(*a, **k)
        return t.double() if _FP64["on"] else t
    return wrapped


# closed-form fp32 inputs; promoted (exactly) to float64 inside ``fp64_mode``
image_batch, sym_uniform, unit_uniform = (_maybe_double(f) for f in (_dw.image_batch, _dw.sym_uniform, _dw.unit_uniform))


@contextlib.contextmanager
def fp64_mode():
    """Run a case in float64: modules are built with double parameters (same closed-form values),
    inputs are promoted.  Used for noise-free CPU references (shared-mask gradient check,
    tools/make_fp32_yardstick.py)."""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    _FP64["on"] = True
    try:
        yield
    finally:
        _FP64["on"] = False
        torch.set_default_dtype(prev)


def _cpu(t):
    return t.detach().float().cpu().contiguous()


def _grads(module, out, tag):
    for name, p in module.named_parameters():
        if p.grad is not None:
            out[f"g/{name}"] = _cpu(p.grad)


def _run_op(module, inputs, tag, device, call=None, wrap=None, sub=False):
    """Forward + backward of one module under the fixed linear loss sum(y * r).  ``sub``: full-size cases (the
    shapes the big tiles are selected at) commit a strided subsample + moments of every tensor (``_sub``)."""
    module = module.to(device)
    target = wrap(module) if wrap else module
    xs = [x.to(device).requires_grad_(True) for x in inputs]
    y = call(target, *xs) if call else target(*xs)
    r = sym_uniform(f"{tag}/r", y.shape).to(device)
    (y.float() * r).sum().backward()
    if sub:
        out = OrderedDict()
        _put(out, "y", y)
        for i, x in enumerate(xs):
            _put(out, f"gx{i}", x.grad)
        for name, p in module.named_parameters():
            if p.grad is not None:
                _put(out, f"g/{name}", p.grad)
        return out
    out = OrderedDict(y=_cpu(y))
    for i, x in enumerate(xs):
        out[f"gx{i}"] = _cpu(x.grad)
    _grads(module, out, tag)
    return out


# ------------------------------------------------------------------------------ op cases


def case_conv(ns, device, *, tag, cin, cout, k, pad, bias, n, h, w, reflect=0):
    m = ns.conv(cin, cout, k, pad, bias)
    fill_state_dict(m, tag)
    x = image_batch(f"{tag}/x", (n, cin, h, w))
    if reflect:
        if hasattr(ns, "conv_reflect"):  # padding fused into the conv loader
            return _run_op(m, [x], tag, device, wrap=lambda mod: ns.conv_reflect(mod, reflect))
        call = lambda mod, t: mod(torch.nn.functional.pad(t, (reflect,) * 4, mode="reflect"))  # noqa: E731
        return _run_op(m, [x], tag, device, call)
    return _run_op(m, [x], tag, device)


def case_modconv(ns, device, *, tag, cin, cout, k, pad, wdim, n, h, w, reflect=0):
    m = ns.modconv(cin, cout, k, wdim, pad)
    fill_state_dict(m, tag)
    x = image_batch(f"{tag}/x", (n, cin, h, w))
    s = unit_uniform(f"{tag}/w", (n, wdim))  # mapping-net outputs are >= 0
    if reflect:
        if hasattr(ns, "modconv_reflect"):
            return _run_op(m, [x, s], tag, device, wrap=lambda mod: ns.modconv_reflect(mod, reflect))
        call = lambda mod, t, wv: mod(torch.nn.functional.pad(t, (reflect,) * 4, mode="reflect"), wv)  # noqa: E731
        return _run_op(m, [x, s], tag, device, call)
    return _run_op(m, [x, s], tag, device)


def case_resample(ns, device, *, tag, kind, n, c, h, w):
    m = {"up": ns.up, "down": ns.down, "blur": ns.smooth}[kind]()
    x = image_batch(f"{tag}/x", (n, c, h, w))
    return _run_op(m, [x], tag, device)


def case_resblock(ns, device, *, tag, dim, n, h, w, sub=False):
    m = ns.resblock(dim)
    fill_state_dict(m, tag)
    return _run_op(m, [image_batch(f"{tag}/x", (n, dim, h, w))], tag, device, sub=sub)


def case_modresblock(ns, device, *, tag, dim, wdim, n, h, w, sub=False):
    m = ns.modresblock(dim, wdim)
    fill_state_dict(m, tag)
    x = image_batch(f"{tag}/x", (n, dim, h, w))
    s = unit_uniform(f"{tag}/w", (n, wdim))
    return _run_op(m, [x, s], tag, device, sub=sub)


# ----------------------------------------------------------------------------- net cases


def _sub(t, limit=4096):
    """Strided subsample + moments for maps too big to commit whole."""
    flat = _cpu(t).flatten()
    if flat.numel() <= limit:
        return {"": flat.reshape(t.shape)}
    step = flat.numel() // limit
    return {
        "/sub": flat[::step][:limit].clone(),
        "/sum": flat.double().sum().float().reshape(1),
        "/sqsum": flat.double().square().sum().float().reshape(1),
    }


def _put(out, key, t):
    for suffix, v in _sub(t).items():
        out[key + suffix] = v


def case_generator(ns, device, *, tag, nc, size, min_latent, n_res, start_filters, n, wdim=6):
    g = ns.Generator(nc, wdim, (size, size), min_latent, n_res, start_filters)
    fill_state_dict(g, tag)
    g = g.to(device)
    x = image_batch(f"{tag}/x", (n, nc, size, size)).to(device).requires_grad_(True)
    w = unit_uniform(f"{tag}/w", (g.n_style_blocks, n, wdim)).to(device).requires_grad_(True)
    z = g.encode(x)
    img = g.decode(z, w)
    feats = g.extract(z, w)
    out = OrderedDict()
    _put(out, "z", z)
    _put(out, "img", img)
    for i, f in enumerate(feats):
        _put(out, f"feat{i}", f)
    r = sym_uniform(f"{tag}/r", img.shape).to(device)
    loss = (img.float() * r).sum() + sum((f.float() ** 2).mean() for f in feats)
    loss.backward()
    _put(out, "gx", x.grad)
    out["gw"] = _cpu(w.grad)
    for name, p in g.named_parameters():
        # biases ahead of InstanceNorm get rounding-noise gradients: skip
        if name in ("encoder.1.bias", "encoder.4.bias", "encoder.8.bias", "encoder.12.bias"):
            continue
        _put(out, f"g/{name}", p.grad)
    return out


def case_patchnet(ns, device, *, tag, kind, nc, size, n, wdim=6):
    net = ns.Discriminator(nc) if kind == "D" else ns.StyleExtractor(nc, wdim)
    fill_state_dict(net, tag)
    net = net.to(device)
    x = image_batch(f"{tag}/x", (n, nc, size, size)).to(device).requires_grad_(True)
    y = net(x)
    r = sym_uniform(f"{tag}/r", y.shape).to(device)
    (y.float() * r).sum().backward()
    out = OrderedDict()
    _put(out, "y", y)
    _put(out, "gx", x.grad)
    for name, p in net.named_parameters():
        if name in ("model.3.bias", "model.7.bias", "model.11.bias"):
            continue
        _put(out, f"g/{name}", p.grad)
    return out


def case_mapping(ns, device, *, tag, n=5, wdim=6, blocks=6):
    m = ns.MappingNetwork(wdim, 2, 0.9)
    fill_state_dict(m, tag)
    m = m.to(device)
    out = OrderedDict()
    z = sym_uniform(f"{tag}/z", (n, wdim)).to(device)
    out["fwd"] = _cpu(m(z))
    torch.manual_seed(77)
    for i in range(4):  # exercises both mixing branches and the CPU-RNG draw order
        out[f"single{i}"] = _cpu(m.get_single_w(n, blocks, device, 1))
    out["single_zero"] = _cpu(m.get_single_w(n, blocks, device, 0))
    d1 = unit_uniform(f"{tag}/d1", (n,)).to(device)
    d2 = unit_uniform(f"{tag}/d2", (n,)).to(device)
    w1, w2 = m.get_two_w(n, blocks, device, (d1, d2))
    out["two_a"], out["two_b"] = _cpu(w1), _cpu(w2)
    out["nomix"] = _cpu(m.get_single_w(n, blocks, device, 0.5, mix_styles=False))
    return out


# ---------------------------------------------------------------------------- loss cases


def case_losses(ns, device, *, tag):
    out = OrderedDict()
    a = sym_uniform(f"{tag}/a", (4, 6)).to(device).requires_grad_(True)
    b = sym_uniform(f"{tag}/b", (4, 6)).to(device).requires_grad_(True)
    l = ns.style_cycle_loss_func(a, b)
    l.backward()
    out["style"], out["style_ga"], out["style_gb"] = _cpu(l).reshape(1), _cpu(a.grad), _cpu(b.grad)

    lat = (sym_uniform(f"{tag}/lat", (4, 8, 6, 6)) * 1.3 + 0.2).to(device).requires_grad_(True)
    l = ns.kl_loss_func(lat)
    l.backward()
    out["kl"], out["kl_g"] = _cpu(l).reshape(1), _cpu(lat.grad)

    f1 = [sym_uniform(f"{tag}/f1_{i}", (3, 4, 5 + i, 5 + i)).to(device).requires_grad_(True) for i in range(3)]
    f2 = [sym_uniform(f"{tag}/f2_{i}", (3, 4, 5 + i, 5 + i)).to(device).requires_grad_(True) for i in range(3)]
    h = (unit_uniform(f"{tag}/h", (3,)) * 0.1 + 0.1).to(device)
    l = ns.path_loss_func(f1, f2, h)
    l.backward()
    out["path"] = _cpu(l).reshape(1)
    for i in range(3):
        out[f"path_g1_{i}"], out[f"path_g2_{i}"] = _cpu(f1[i].grad), _cpu(f2[i].grad)
    return out


# ---------------------------------------------------------------- host state-machine cases


def case_adap(ns, device, *, tag):
    ctl = ns.ADAp(256, 5.12e-4, 4, 0.6)
    scores = unit_uniform(f"{tag}/scores", (200,)) * 2 - 1
    scores[60:140] = scores[60:140].abs() * 0.3 + 0.7  # a stretch above the 0.6 target
    trace = []
    for v in scores:
        ctl.update_p(v.clone())
        trace.append(ctl())
    return OrderedDict(trace=torch.tensor(trace))


def case_imagebuffer(ns, device, *, tag):
    random.seed(42)
    buf = ns.ImageBuffer(6)
    sums = []
    for step in range(8):
        imgs = (torch.arange(4, dtype=torch.float32) + 4 * step).view(4, 1, 1, 1).expand(4, 1, 2, 2).to(device)
        sums.append(_cpu(buf(imgs))[:, 0, 0, 0])
    return OrderedDict(ids=torch.stack(sums))


# ---------------------------------------------------------------------------- step cases


def make_config(nc, size, batch, min_latent=64, lr=2e-3):
    """Stock config.toml values with image_size / image_channels / batch_size (and, for the
    3-downsample case, min_latent_resolution and the learning rate) overridden."""
    return {
        "training": {"batch_size": batch, "random_seed": 42, "training_steps": 150000,
                     "image_buffer_size": 100, "style_mixing_prob": 0.9,
                     "deterministic_cuda_kernels": False, "gpu_number": 0},
        "optimisation": {"style_cycle_loss_lambda": 5.0, "identity_loss_lambda": 5.0,
                         "reconstruction_loss_lambda": 5.0, "kl_loss_lambda": 0.01,
                         "path_loss_lambda": 0.1, "path_loss_jacobian_granularity": [0.1, 0.2],
                         "learning_rate": lr, "mapping_network_learning_rate": 2e-5,
                         "adam_betas": [0.5, 0.99]},
        "ada": {"discriminator_real_acc_target": 0.6,
                "ada_overfitting_measurement_n_images": 256, "ada_adjustment_size": 5.12e-4},
        "architecture": {"w_dim": 6, "add_latent_noise": False, "min_latent_resolution": min_latent,
                         "n_resnet_blocks": 7, "mapping_network_layers": 2},
        "data": {"image_size": list(size), "image_channels": nc},
    }


@contextlib.contextmanager
def cpu_uniform_():
    """Route Tensor.uniform_ through the CPU generator so the device-RNG draw of ``h``
    (training.py:216-223) is identical on CPU and GPU runs."""
    orig = torch.Tensor.uniform_

    def patched(self, a=0.0, b=1.0, **kw):
        if self.device.type == "cpu":
            return orig(self, a, b, **kw)
        tmp = orig(torch.empty(self.shape, dtype=torch.float32), a, b)
        return self.copy_(tmp)

    torch.Tensor.uniform_ = patched
    try:
        yield
    finally:
        torch.Tensor.uniform_ = orig


def build_step_state(ns, device, cfg, tag):
    a, t, d = cfg["architecture"], cfg["training"], cfg["data"]
    nets = {
        "D": ns.Discriminator(d["image_channels"]),
        "G": ns.Generator(d["image_channels"], a["w_dim"], tuple(d["image_size"]),
                          a["min_latent_resolution"], a["n_resnet_blocks"]),
        "M": ns.MappingNetwork(a["w_dim"], a["mapping_network_layers"], t["style_mixing_prob"]),
        "S": ns.StyleExtractor(d["image_channels"], a["w_dim"]),
    }
    for k, n in nets.items():
        fill_state_dict(n, f"{tag}/{k}", bias_scale=0.0 if k != "M" else 0.1)
        nets[k] = n.to(device)
    o = cfg["optimisation"]
    mk = getattr(ns, "make_adam", None) or (
        lambda net, lr, betas: torch.optim.Adam(net.parameters(), lr=lr, betas=betas))
    opts = {k: mk(nets[k], o["mapping_network_learning_rate"] if k == "M" else o["learning_rate"],
                  tuple(o["adam_betas"])) for k in nets}
    return nets, opts


def _batches(tag, stream, cfg):
    d, b = cfg["data"], cfg["training"]["batch_size"]
    i = 0
    while True:
        yield image_batch(f"{tag}/{stream}{i}", (b, d["image_channels"], *d["image_size"]))
        i += 1


def case_steps(ns, device, *, tag, nc, size, batch, n_steps=2, seed=1234, min_latent=64, lr=2e-3):
    cfg = make_config(nc, size, batch, min_latent, lr)
    nets, opts = build_step_state(ns, device, cfg, tag)
    prints, marks = _batches(tag, "print", cfg), _batches(tag, "mark", cfg)
    buf = ns.ImageBuffer(cfg["training"]["image_buffer_size"])
    ada = ns.make_ada().to(device)
    ada_p = ns.ADAp(cfg["ada"]["ada_overfitting_measurement_n_images"], cfg["ada"]["ada_adjustment_size"],
                    batch, cfg["ada"]["discriminator_real_acc_target"])
    torch.manual_seed(seed)
    random.seed(seed)
    out = OrderedDict()
    with cpu_uniform_():
        for s in range(n_steps):
            ada.set_p(ada_p())
            dl, (ra, fa) = ns.discriminator_step(cfg, device, nets["D"], nets["G"], nets["M"], opts["D"],
                                                 prints, marks, buf, ada, ada_p)
            gl, parts = ns.generator_step(cfg, device, nets["G"], nets["D"], nets["M"], nets["S"],
                                          opts["G"], opts["M"], opts["S"], prints, marks, ada)
            out[f"step{s}/d"] = torch.tensor([float(v) for v in (dl, ra, fa)])  # (float(): LoggedScalar when async)
            out[f"step{s}/g"] = torch.tensor([float(v) for v in (gl, *parts)])
    # post-step probes: network outputs on a fixed input are a smooth function of the
    # updated parameters (raw parameter checksums are dominated by sign(g) of tiny grads)
    with torch.no_grad():
        x = image_batch(f"{tag}/probe", (2, nc, *size)).to(device)
        w = unit_uniform(f"{tag}/probe_w", (nets["G"].n_style_blocks, 2, cfg["architecture"]["w_dim"])).to(device)
        _put(out, "probe/img", nets["G"](x, w))
        _put(out, "probe/d", nets["D"](x))
        out["probe/s"] = _cpu(nets["S"](x))
        out["probe/m"] = _cpu(nets["M"](sym_uniform(f"{tag}/probe_z", (3, cfg["architecture"]["w_dim"])).to(device)))
    return out


# -------------------------------------------------------------------------------- registry

CASES = OrderedDict()


def _reg(name, fn, **kw):
    CASES[name] = (fn, dict(tag=name, **kw))


_reg("conv3_p1", case_conv, cin=8, cout=16, k=3, pad=1, bias=True, n=2, h=9, w=11)
_reg("conv3_reflect", case_conv, cin=16, cout=8, k=3, pad=0, bias=False, n=2, h=10, w=9, reflect=1)
_reg("conv4_p1", case_conv, cin=8, cout=8, k=4, pad=1, bias=True, n=2, h=13, w=12)
_reg("conv4_rgb", case_conv, cin=3, cout=16, k=4, pad=1, bias=True, n=2, h=17, w=16)
_reg("conv7_reflect_rgb", case_conv, cin=3, cout=8, k=7, pad=0, bias=True, n=2, h=12, w=14, reflect=3)
_reg("conv7_tail", case_conv, cin=16, cout=3, k=7, pad=0, bias=True, n=2, h=12, w=12, reflect=3)
_reg("conv7_tail64", case_conv, cin=64, cout=3, k=7, pad=0, bias=True, n=2, h=16, w=20, reflect=3)  # s2d path
_reg("conv4_head", case_conv, cin=32, cout=1, k=4, pad=1, bias=True, n=2, h=9, w=9)
_reg("modconv_p1", case_modconv, cin=16, cout=8, k=3, pad=1, wdim=6, n=3, h=10, w=12)
_reg("modconv_reflect", case_modconv, cin=8, cout=8, k=3, pad=0, wdim=6, n=2, h=9, w=9, reflect=1)
# 128-channel layers: the smallest shapes the fp8 path (config #5: Ci % 128 == 0) applies to
_reg("conv3_c128", case_conv, cin=128, cout=128, k=3, pad=1, bias=True, n=2, h=12, w=12)
_reg("modconv_c128", case_modconv, cin=128, cout=128, k=3, pad=0, wdim=6, n=2, h=16, w=16, reflect=1)
_reg("blur_even", case_resample, kind="blur", n=2, c=8, h=8, w=10)
_reg("up_even", case_resample, kind="up", n=2, c=8, h=8, w=6)
_reg("up_odd", case_resample, kind="up", n=1, c=8, h=7, w=9)
_reg("down_even", case_resample, kind="down", n=2, c=8, h=12, w=16)
_reg("down_odd", case_resample, kind="down", n=2, c=8, h=15, w=31)
_reg("down_odd2", case_resample, kind="down", n=1, c=16, h=63, w=9)
_reg("resblock", case_resblock, dim=8, n=2, h=9, w=10)
_reg("modresblock", case_modresblock, dim=8, wdim=6, n=2, h=9, w=9)
_reg("resblock_c128", case_resblock, dim=128, n=1, h=16, w=16)
# The residual blocks of the 256 x 256 step AT THEIR REAL SIZE (256 channels, 64 x 64 latent, B = 16 = 256 tiles of
# 256 x 256): the shapes at which the phase-pipelined igemm (InstanceNorm-partial epilogue, reflect fold, residual
# gradient through BlockLink), the phase-pipelined weight gradient and the batched style path are SELECTED -- every
# smaller case runs the generic tiles.  ~10 s each on the CPU oracle; outputs subsampled (tests/cases.py::_sub).
_reg("resblock_c256_b16", case_resblock, dim=256, n=16, h=64, w=64, sub=True)
_reg("modresblock_c256_b16", case_modresblock, dim=256, wdim=6, n=16, h=64, w=64, sub=True)
_reg("gen32", case_generator, nc=3, size=32, min_latent=8, n_res=3, start_filters=8, n=2)
_reg("gen64_gray", case_generator, nc=1, size=64, min_latent=64, n_res=7, start_filters=16, n=1)
# BASELINE config #4's topology (512x512: 3 downsamples, 512-channel latent) at a size the CPU finishes
_reg("gen64_deep", case_generator, nc=3, size=64, min_latent=8, n_res=3, start_filters=64, n=1)
_reg("disc32", case_patchnet, kind="D", nc=3, size=32, n=2)
_reg("disc64", case_patchnet, kind="D", nc=3, size=64, n=1)
_reg("style32", case_patchnet, kind="S", nc=3, size=32, n=2)
_reg("style64_gray", case_patchnet, kind="S", nc=1, size=64, n=2)
_reg("mapping", case_mapping)
_reg("losses", case_losses)
_reg("adap", case_adap)
_reg("imagebuffer", case_imagebuffer)
_reg("steps64", case_steps, nc=1, size=(64, 64), batch=4)        # BASELINE config #1
_reg("steps256", case_steps, nc=3, size=(256, 256), batch=2)     # north-star shape, B=2
# config #4's topology through the whole step: 3 downsamples, 512-channel latent at 16x16, Co > 256 tiles.
# Learning rate 2e-5: Adam's first update is lr * sign(g) per weight, so at the stock 2e-3 the post-step
# probes of this 36 M-parameter generator mostly measure which way the noise-level gradients happened to
# round (fp32 mode 6.6e-2, bf16 0.49 on probe/img -- measured); at 2e-5 they test the step itself.
_reg("steps128", case_steps, nc=3, size=(128, 128), batch=2, min_latent=16, lr=2e-5)
# ... and the same two steps at the STOCK learning rate (config.toml: 2e-3), so that config #4's topology is also
# held to the reference on updates of the real size: the logged losses of both steps are compared at the usual
# bounds; the post-step probes get the bounds measured for this case (tests/test_hip_parity.py::PROBE_TOL).
_reg("steps128_stock", case_steps, nc=3, size=(128, 128), batch=2, min_latent=16, lr=2e-3)

SLOW_CASES = {"steps256", "steps128", "steps128_stock"}


def run_case(name, ns, device):
    fn, kw = CASES[name]
    return fn(ns, torch.device(device), **kw)


def run_case_shared_masks(name, product_namespace, oracle_namespace):
    """The net-level gradient check without activation-mask noise.

    A forward error eps flips the ReLU / LeakyReLU mask of the ~eps fraction of pre-activations that
    sit within eps of zero, and every flip is an O(1) change of that element's gradient -- noise that
    says nothing about the kernels.  Here the HIP path runs first and records the sign mask of every
    fused activation (ops.ACT_TAP); the oracle then runs in FLOAT64 with those masks REPLAYED in place
    of its own ReLU / LeakyReLU decisions (oracle.model.ACT_OVERRIDE).  Both sides differentiate the
    same piecewise-linear function, so what remains is kernel error (times the conditioning of the
    case).  Returns (product outputs, oracle outputs, flipped mask elements, mask elements)."""
    import one_to_many_gan_amd.ops as pops
    import oracle.model as om

    tape = []
    pops.ACT_TAP = tape
    try:
        got = run_case(name, product_namespace, "cuda")
    finally:
        pops.ACT_TAP = None
    masks = iter(tape)
    count = [0, 0]

    def replay(kind, t, inplace):
        # .contiguous(): the tap hands out NHWC-strided views; with them torch's CPU kernels switch the
        # oracle to channels-last tensors, whose float64 path returned wrong gradients at batch 1
        m = next(masks)[:, : t.shape[1]].contiguous()
        assert m.shape == t.shape, (kind, tuple(m.shape), tuple(t.shape))
        count[0] += int(((t.detach() > 0) != m).sum())
        count[1] += m.numel()
        if kind == "relu":
            return t.mul_(m.to(t.dtype)) if inplace else t * m.to(t.dtype)
        return torch.where(m, t, 0.2 * t)

    om.ACT_OVERRIDE = replay
    torch.set_num_threads(host_threads())
    try:
        with fp64_mode():
            want = run_case(name, oracle_namespace, "cpu")
    finally:
        om.ACT_OVERRIDE = None
    assert next(masks, None) is None, "the oracle applied fewer activations than the HIP path recorded"
    return got, want, count[0], count[1]
