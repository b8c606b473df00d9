"""CPU tests of the operator surface (csrc/torch_ops.cpp): every launcher of include/o2m_hip.h is a
dispatcher-visible ``torch.ops.o2m.*`` op with a schema, CUDA + Meta kernels, argument checks that
raise RuntimeError, and no CPU fallback.  (The compute itself is covered by the -m gpu suites, which
reach the kernels through these same ops.)"""

import pytest
import torch

from one_to_many_gan_amd import _hip


def _meta(*shape, dtype=torch.bfloat16):
    return torch.empty(*shape, device="meta", dtype=dtype)


def test_every_launcher_has_a_schema_with_mutable_outputs():
    ops = _hip.ops()
    for name in set(_hip.SIGNATURES) - _hip.MEASUREMENT_ONLY:
        op = getattr(ops, name[len("o2m_"):]).default
        assert op._schema.name == "o2m::" + name[len("o2m_"):]
    s = ops.conv2d_fwd.default._schema
    assert [a.name for a in s.arguments][:3] == ["x", "w", "y"]
    assert s.arguments[2].alias_info is not None and s.arguments[2].alias_info.is_write  # y is written
    assert ops.adam_step.default._schema.arguments[0].alias_info.is_write
    assert len(s.returns) == 0


def test_meta_kernels_check_shapes_and_cpu_is_rejected():
    ops = _hip.ops()
    x, w, y = _meta(2, 8, 8, 16), _meta(32, 3, 3, 16), _meta(2, 8, 8, 32)
    ops.conv2d_fwd(x, w, y, None, None, None, None, 1, _hip.PAD_ZERO, _hip.ACT_RELU, False, 1)
    with pytest.raises(RuntimeError, match=r"y must be \[2, 8, 8, 32\]"):
        ops.conv2d_fwd(x, w, _meta(2, 8, 8, 24), None, None, None, None, 1, 0, 1, False, 1)
    with pytest.raises(RuntimeError, match="share one dtype"):
        ops.conv2d_fwd(x, _meta(32, 3, 3, 16, dtype=torch.float32), y, None, None, None, None, 1, 0, 1, False, 1)
    with pytest.raises(RuntimeError, match="must be float32"):
        ops.conv2d_fwd(x, w, y, None, None, _meta(32), None, 1, 0, 1, False, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d_fwd(torch.zeros(2, 8, 8, 16), torch.zeros(32, 3, 3, 16), torch.zeros(2, 8, 8, 32), None, None,
                       None, None, 1, 0, 1, False, 1)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.instnorm_apply(_meta(2, 8, 8, 16).permute(0, 2, 1, 3), _meta(2, 16, 2, dtype=torch.float32), None,
                           _meta(2, 8, 8, 16), 0)
    dw = _meta(32, 3, 3, 16, dtype=torch.float32)
    ops.conv2d_wgrad(x, y, dw, None, None, 1, 0, 0, 1)
    with pytest.raises(RuntimeError, match="shapes of x / gy / dw disagree"):
        ops.conv2d_wgrad(_meta(2, 8, 8, 8), y, dw, None, None, 1, 0, 0, 1)
    with pytest.raises(RuntimeError, match="workspace too small"):
        ops.instnorm_stats(x, _meta(1, dtype=torch.float32), _meta(2, 16, 2, dtype=torch.float32), 1e-5)
    with pytest.raises(RuntimeError, match="dtype must be bfloat16 or float32"):
        ops.pack_nchw(_meta(1, 3, 4, 4, dtype=torch.float32), _meta(1, 4, 4, 8, dtype=torch.float16))


def test_ops_trace_under_fake_tensor_mode():
    """FakeTensor tracing (what torch.compile / hipGraph capture planning needs) sees the ops."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    ops = _hip.ops()
    with FakeTensorMode():
        x = torch.empty(2, 8, 8, 16, device="cuda", dtype=torch.bfloat16)
        w = torch.empty(32, 3, 3, 16, device="cuda", dtype=torch.bfloat16)
        y = torch.empty(2, 8, 8, 32, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_fwd(x, w, y, None, None, None, None, 1, 0, 1, False, 1)
        mr = torch.empty(2, 32, 2, device="cuda")
        part = torch.empty(ops.instnorm_ws_floats(2, 64, 32), device="cuda")
        ops.instnorm_stats(y, part, mr, 1e-5)
        ops.instnorm_apply(y, mr, None, torch.empty_like(y), _hip.ACT_RELU)
        p = torch.empty(128, device="cuda")
        ops.adam_step(p, p.clone(), p.clone(), p.clone(), torch.empty(1, device="cuda"), 1e-3, 0.5, 0.99, 1e-8, 1.0)


def test_every_launch_goes_through_the_device_guard_and_train_selects_its_device():
    """with gpu_number != 0 a launch on torch's default stream (handle 0) would bind to device 0
    while its pointers live on device N.  Host-side check of the two places that prevent it: every
    launcher call in the operator shim sits inside O2M_CALL (which builds `Launch`: c10::DeviceGuard on
    the tensor's device + that device's current stream), and train.py selects its device before building
    anything."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "one_to_many_gan_amd", "csrc", "torch_ops.cpp")).read()
    assert re.search(r"struct Launch \{[^}]*c10::DeviceGuard guard;", shim, re.S)
    assert "getCurrentHIPStreamMasqueradingAsCUDA" in shim and "Launch L(t);" in shim
    # launcher symbols that take a stream appear nowhere outside an O2M_CALL(...) argument
    launchers = [n for n, (_, a) in _hip.SIGNATURES.items() if a and a[-1] is _hip._vp and n not in _hip.MEASUREMENT_ONLY]
    assert len(launchers) >= 25
    body = shim[shim.index("#define O2M_CALL"):]
    for name in launchers:
        for m in re.finditer(r"\b" + name + r"\(", body):
            before = body[max(0, m.start() - 200): m.start()]
            assert "O2M_CALL(" in before, name
    train = open(os.path.join(root, "train.py")).read()
    run_src = train[train.index("def run("):]
    assert run_src.index("torch.cuda.set_device(device)") < run_src.index("Generator(")


@pytest.mark.gpu
def test_functional_ops_are_registered_with_autograd():
    """torch.ops.o2m.resample / pair_sum / moments: functional operators of the o2m namespace whose derivative is attached
    with torch.library.register_autograd -- they differentiate through the dispatcher like ATen
    ops, have fake kernels (shape inference on meta tensors), pass torch.library.opcheck, and agree with the torch
    formulas they stand for (loss.py:82-111, layers.py:191-247)."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H
    from one_to_many_gan_amd import ops

    torch.manual_seed(3)
    dev = "cuda"
    # ---- losses
    a = torch.randn(4, 6, 10, 16, device=dev, requires_grad=True)
    b = torch.randn(4, 6, 10, 16, device=dev, requires_grad=True)
    w = torch.rand(4, device=dev) + 0.5
    for mode, ref in ((H.RED_L1, lambda: (a - b).abs().sum()),
                      (H.RED_SQ, lambda: (w.view(4, 1, 1, 1) * (a - b) ** 2).sum())):
        got = torch.ops.o2m.pair_sum(a, b, w if mode == H.RED_SQ else None, mode)
        ga, gb = torch.autograd.grad(got * 0.7, (a, b))
        ra, rb = torch.autograd.grad(ref() * 0.7, (a, b))
        assert abs(float(got) - float(ref())) <= 1e-4 * abs(float(ref()))
        assert torch.allclose(ga, ra, rtol=1e-5, atol=1e-6) and torch.allclose(gb, rb, rtol=1e-5, atol=1e-6)
    s1, s2 = torch.ops.o2m.moments(a)
    (g,) = torch.autograd.grad(2.0 * s1 + 0.5 * s2, a)
    assert torch.allclose(g, 2.0 + a.detach(), rtol=1e-5, atol=1e-6)
    (g,) = torch.autograd.grad(torch.ops.o2m.moments(a)[1], a)   # one output unused
    assert torch.allclose(g, 2.0 * a.detach(), rtol=1e-5, atol=1e-6)
    # ---- resample: the adjoint pair (<D x, g> = <x, D^T g>) through autograd, for all three operators
    x = torch.randn(2, 12, 20, 16, device=dev, requires_grad=True)
    for kind in ("blur", "up", "down"):
        y = ops.resample(x, kind)
        gy = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, gy)
        lhs, rhs = float((y.detach().double() * gy.double()).sum()), float((x.detach().double() * gx.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0), kind
        assert torch.ops.o2m.resample(torch.empty(2, 12, 20, 16, device="meta"), kind).shape == y.shape
    ref = F.interpolate(x.detach().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
    assert ops.resample(x, "up").shape == (2, 24, 40, 16) and ref.shape == (2, 16, 24, 40)
    # ---- the operator checker: schema, fake kernel, autograd registration
    torch.library.opcheck(torch.ops.o2m.pair_sum.default, (a.detach().requires_grad_(True), b.detach(), w, H.RED_SQ),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.o2m.resample.default, (x.detach().requires_grad_(True), "down"),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.o2m.moments.default, (a.detach().requires_grad_(True),),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
