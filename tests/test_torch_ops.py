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
