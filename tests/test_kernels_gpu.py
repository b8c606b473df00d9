"""Direct kernel checks (-m gpu), through the o2m:: operator shim and the C ABI behind it, for
entry-point features and full-size shapes the network-level parity cases do not reach."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_wgrad_slab_workspaces_are_per_stream_and_survive_a_regrow():
    """VERDICT r2 #13 / ADVICE: the slab workspace used to be ONE buffer per device whatever the stream, so
    weight-gradient launches alternating between the main stream and the weight-gradient stream wrote the same
    slabs concurrently, and a regrow freed a block a kernel on the other stream still used.  The workspaces are
    now keyed by (device, stream).  Launches alternate between two streams WITHOUT synchronisation in between,
    the second round needs a larger workspace than the first (regrow on both streams); every result is checked
    against an fp64 reference and against a quiet single-stream run bit for bit."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(5)
    dev = torch.device("cuda")
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)

    def problem(B, S, Ci, Co):
        x = torch.randn(B, S, S, Ci, device=dev).bfloat16()
        g = torch.randn(B, S, S, Co, device=dev).bfloat16()
        return x, g

    def reference(x, g):
        xp = torch.nn.functional.pad(x.double().cpu().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
        co, ci = g.shape[-1], x.shape[-1]
        return torch.nn.grad.conv2d_weight(xp, (co, ci, 3, 3), g.double().cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)

    small = [problem(2, 32, 64, 64) for _ in range(4)]       # ~0.3 M slab floats per launch
    large = [problem(4, 64, 256, 256) for _ in range(4)]     # ~12 M: both streams' workspaces regrow
    saved = H._SLABS
    H._SLABS = H._StreamScratch(1 << 12)  # (the production floor of 16 M floats would hide the regrow)
    outs = []
    torch.cuda.synchronize()
    try:
        for probs in (small, large):
            for k, (x, g) in enumerate(probs):
                st = side if k % 2 else main
                dw = torch.zeros(g.shape[-1], 3, 3, x.shape[-1], device=dev)
                if st is side:
                    side.wait_stream(main)  # operands / zero fill were produced on the main stream
                with torch.cuda.stream(st):
                    H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_REFLECT)
                outs.append((x, g, dw))
        torch.cuda.synchronize()
        keys = {k[1] for k in H._SLABS.buf}
        assert len(keys) == 2, "one slab workspace per stream"
        assert all(ws.numel() > (1 << 20) for ws in H._SLABS.buf.values()), "both workspaces regrew"
    finally:
        H._SLABS = saved
    for x, g, dw in outs:
        ref = reference(x, g)
        assert ((dw.double().cpu() - ref).norm() / ref.norm()) < 1e-5
        quiet = torch.zeros_like(dw)
        H.conv2d_wgrad(x, g, quiet, pad=1, pad_mode=H.PAD_REFLECT)
        torch.cuda.synchronize()
        assert torch.equal(quiet, dw), "a launch beside another stream's launch changed the result"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_wgrad_is_bitwise_reproducible(dt):
    """Slab mode (default): every pixel slice stores its partial and a second kernel sums the slices
    in slice order, so two launches on the same operands agree bit for bit -- the property the
    reference's deterministic_cuda_kernels switch asks for (train.py:41-45).  The atomic form
    (O2M_WGRAD_ATOMICS=1) differs in the last bits from run to run."""
    from one_to_many_gan_amd import _hip as H

    assert not H.WGRAD_ATOMICS
    torch.manual_seed(2)
    B, S, Ci, Co = 16, 64, 256, 256
    x = torch.randn(B, S, S, Ci, device="cuda").to(dt)
    g = torch.randn(B, S, S, Co, device="cuda").to(dt)
    outs = []
    for _ in range(3):
        dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_REFLECT)
        outs.append(dw)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # accumulation semantics: a second launch ADDS to dw
    H.conv2d_wgrad(x, g, outs[0], pad=1, pad_mode=H.PAD_REFLECT)
    assert float((outs[0] - 2 * outs[1]).abs().max()) <= 1e-6 * float(outs[1].abs().max())


def test_strided_conv_matches_torch():
    """o2m_conv_desc.stride (used by the space-to-depth form of the RGB tail conv)."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(1)
    B, Hh, Ww, Ci, Co, k, s = 2, 16, 24, 64, 16, 6, 4
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / 48).to(torch.bfloat16)
    ho, wo = (Hh + 2 - k) // s + 1, (Ww + 2 - k) // s + 1
    y = torch.empty(B, ho, wo, Co, device="cuda", dtype=torch.bfloat16)
    H.conv2d_fwd(x, w, y, pad=1, pad_mode=H.PAD_REFLECT, act=H.ACT_NONE, stride=s)
    xin = torch.nn.functional.pad(x.float().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
    ref = torch.nn.functional.conv2d(xin, w.float().permute(0, 3, 1, 2), stride=s).permute(0, 2, 3, 1)
    assert ((y.float() - ref).norm() / ref.norm()) < 5e-3


ADJOINT_SHAPES = [
    # B, H, W, Ci, Co, k, pad, reflect                      tile / path it reaches (bf16)
    (16, 64, 64, 256, 256, 3, 1, True),      # config #2 latent layer: igemm 256x256, wgrad co128xk128
    (16, 256, 256, 64, 128, 3, 1, False),    # igemm 256x64 (short reduction), wgrad co128xk256
    (16, 256, 256, 128, 64, 3, 1, False),    # igemm 256x64 forward, 256x128 data gradient, wgrad co64xk128
    (16, 255, 255, 64, 128, 4, 1, False),    # discriminator: odd sizes, ragged tiles
    (8, 128, 128, 256, 512, 3, 1, False),    # config #4 (512x512, B = 8): Co > 256 -> two N tiles, wgrad co256xk128
    (8, 64, 64, 512, 512, 3, 1, True),       # config #4 latent layer
    (8, 31, 31, 256, 512, 4, 1, False),      # discriminator head trunk: small-M 128x128 tiles
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", ADJOINT_SHAPES, ids=lambda s: "x".join(map(str, s[:7])))
def test_full_size_adjoint_identities(shape, dt):
    """Size-independent properties at BASELINE config #2 / #4 sizes, where the CPU oracle is too slow:
    forward, data-gradient and weight-gradient kernels must be mutually adjoint,
        <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(x, g)>,
    and the forward must be linear in x.  Both precisions: fp32 (bf16x3 split; register staging)
    and bf16 (the LDS-DMA loaders); inner products evaluated in fp64.  In bf16 the STORED y / gx
    are rounded to 8 bits, which bounds the identities at ~2^-9 / sqrt(N) -- far below the 2e-3
    asked here, which a wrong tap, tile edge or channel slice would miss by orders of magnitude."""
    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, k, pad, reflect = shape
    pm = H.PAD_REFLECT if reflect else H.PAD_ZERO
    torch.manual_seed(3)
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").to(dt)
    w = (torch.randn(Co, k, k, Ci, device="cuda") / (Ci * k * k) ** 0.5).to(dt)
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    g = torch.randn(B, ho, wo, Co, device="cuda").to(dt)
    y = torch.empty(B, ho, wo, Co, device="cuda", dtype=dt)
    H.conv2d_fwd(x, w, y, pad=pad, pad_mode=pm, act=H.ACT_NONE)
    lhs = float((y.double() * g.double()).sum())
    # weight gradient
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    H.conv2d_wgrad(x, g, dw, pad=pad, pad_mode=pm)
    via_w = float((dw.double() * w.double()).sum())
    # data gradient = forward kernel with the flipped / transposed filter (+ fold for reflect)
    w_d = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    if reflect:
        gxp = torch.empty(B, Hh + 2 * pad, Ww + 2 * pad, Ci, device="cuda", dtype=dt)
        H.conv2d_fwd(g, w_d, gxp, pad=k - 1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
        gx = torch.empty_like(x)
        H.fold_scale_dot(gxp, None, None, gx, None, pad)
    else:
        gx = torch.empty_like(x)
        H.conv2d_fwd(g, w_d, gx, pad=k - 1 - pad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
    via_x = float((gx.double() * x.double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    tol = 2e-5 if dt == torch.float32 else 2e-3
    assert abs(lhs - via_w) < tol * scale, (lhs, via_w, scale)
    assert abs(lhs - via_x) < tol * scale, (lhs, via_x, scale)
    # linearity: conv(2x) == 2 conv(x) (a power-of-two scale commutes with every rounding step)
    y2 = torch.empty_like(y)
    H.conv2d_fwd((2 * x).contiguous(), w, y2, pad=pad, pad_mode=pm, act=H.ACT_NONE)
    assert float((y2.float() - 2 * y.float()).norm() / y2.float().norm()) < 1e-6


FULL_SIZE_CONVS = [
    # B, H, W, Ci, Co, k, pad, reflect, features              (bf16: which igemm kernel / tile)
    (16, 64, 64, 256, 256, 3, 1, True, "plain"),     # phase-pipelined 256x256 kernel, one tile per CU
    (32, 64, 64, 256, 256, 3, 1, True, "epilogue"),  # same, two waves of tiles; bias + act + residual
    (16, 64, 64, 256, 256, 3, 1, True, "modulated"),  # per-sample filters + demodulation scale (decoder)
    (16, 66, 66, 256, 256, 3, 2, False, "plain"),     # zero padding 2 = the data-gradient call shape
    (8, 64, 64, 512, 512, 3, 1, True, "plain"),       # config #4: two N tiles, 72 K-tiles
    (16, 128, 128, 128, 256, 3, 1, False, "epilogue"),  # Ci = 128: a tap is two K-tiles
    (16, 256, 256, 128, 64, 3, 1, False, "epilogue"),  # N = 64 tile
    (16, 128, 128, 256, 128, 3, 1, False, "modulated"),  # N = 128 tile, modulated
    # round 3: the halo-tile kernel (zero pad 1, Co = 64 / 128, W % 32 == 0, >= 512 tiles) in each of its forms
    (16, 256, 256, 64, 128, 3, 1, False, "plain"),       # Co = 128, one 64-channel chunk
    (4, 256, 256, 128, 64, 3, 1, False, "modulated"),    # Co = 64, two chunks, per-sample filters + demodulation
    (2, 256, 256, 64, 64, 3, 1, False, "epilogue"),      # the minimum grid (512 tiles), bias + act + residual
    # the p8 kernel's per-fill tap masks (border fills: zero fill or mirrored element per tap from a bit mask)
    (20, 63, 63, 128, 256, 4, 1, False, "epilogue"),     # 4 x 4 taps, odd map (discriminator trunk), p8 + 128x128 tail
    (16, 64, 64, 256, 256, 5, 2, True, "plain"),         # reflect pad 2: NOT the p8 kernel's geometry -> symmetric kernel
]


@pytest.mark.parametrize("case", FULL_SIZE_CONVS, ids=lambda c: "x".join(map(str, c[:7])) + "-" + c[8])
def test_full_size_bf16_conv_matches_torch_fp32(case):
    """The bf16 igemm kernels at BASELINE config #2 / #4 layer sizes -- too big for the CPU oracle --
    against torch's own fp32 convolution of the SAME bf16-valued operands on the GPU.  The only
    differences left are the summation order and the final rounding of y to bf16 (2^-9 relative per
    element, ~1.2e-3 rms), so 3e-3 relative L2 catches any mis-staged tile, tap or K-tile."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, k, pad, reflect, feat = case
    torch.manual_seed(17)
    dev, dt = "cuda", torch.bfloat16
    x = torch.randn(B, Hh, Ww, Ci, device=dev).to(dt)
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.empty(B, ho, wo, Co, device=dev, dtype=dt)
    kw = dict(pad=pad, pad_mode=H.PAD_REFLECT if reflect else H.PAD_ZERO, act=H.ACT_NONE)
    w = (torch.randn((B if feat == "modulated" else 1), Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5).to(dt)
    scale = bias = res = None
    if feat == "modulated":
        scale = torch.rand(B, Co, device=dev) + 0.5
        H.conv2d_fwd(x, w, y, out_scale=scale, per_sample_w=True, **kw)
    elif feat == "epilogue":
        bias = torch.randn(Co, device=dev)
        res = torch.randn(B, ho, wo, Co, device=dev).to(dt)
        kw["act"] = H.ACT_LRELU
        H.conv2d_fwd(x, w[0], y, bias=bias, residual=res, **kw)
    else:
        H.conv2d_fwd(x, w[0], y, **kw)
    xin = x.float().permute(0, 3, 1, 2)
    xin = F.pad(xin, (pad,) * 4, mode="reflect") if reflect else F.pad(xin, (pad,) * 4)
    if feat == "modulated":  # grouped conv with per-sample filters, as the reference runs it
        ref = F.conv2d(xin.reshape(1, B * Ci, *xin.shape[2:]), w.float().permute(0, 1, 4, 2, 3).reshape(B * Co, Ci, k, k),
                       groups=B).view(B, Co, ho, wo) * scale.view(B, Co, 1, 1)
    else:
        ref = F.conv2d(xin, w[0].float().permute(0, 3, 1, 2))
    if feat == "epilogue":
        ref = F.leaky_relu(ref + bias.view(1, -1, 1, 1), 0.2) + res.float().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    err = float((y.float() - ref).norm() / ref.norm())
    assert err < 3e-3, err
    worst = float((y.float() - ref).abs().max() / ref.abs().max())
    assert worst < 2e-2, worst  # no single wrong tile hiding in the norm


DIRECT_CONVS = [
    # B, H, W, Ci, Co, k, pad, reflect, extras, expected kernel                       (conv_direct.hip)
    (4, 256, 256, 8, 64, 4, 1, False, "bias+lrelu", "conv_stem8<bf16,4x4>"),   # D / S stem (builder.py:269): 255 x 255 out
    (3, 128, 96, 8, 64, 4, 1, False, "plain", "conv_stem8<bf16,4x4>"),         # rows % 8 != 0, last strip of 31 pixels
    (2, 256, 256, 8, 64, 7, 3, True, "bias+stats", "conv_stem8<bf16,7x7>"),    # G encoder stem behind ReflectionPad2d(3)
    (2, 64, 40, 8, 64, 7, 3, True, "bias+stats", "conv_stem8<bf16,7x7>"),      # narrow map: one strip, 24 lanes masked
    (2, 256, 256, 8, 64, 7, 6, False, "plain", "conv_stem8<bf16,7x7>"),        # data gradient of the 7 x 7 image head: 262 x 262
    (2, 255, 255, 64, 8, 4, 2, False, "plain", "conv_fewout<bf16,4x4>"),       # data gradient of the D / S stem: 256 x 256 out
    (4, 30, 30, 512, 8, 4, 1, False, "bias", "conv_fewout<bf16,4x4>"),         # D head 512 -> 1 (builder.py:284): 29 x 29 out
    (1, 9, 70, 192, 8, 4, 1, False, "bias", "conv_fewout<bf16,4x4>"),          # three 64-channel chunks, tiles clipped on both axes
]


@pytest.mark.parametrize("case", DIRECT_CONVS, ids=lambda c: "x".join(map(str, c[:7])) + "-" + c[8])
def test_direct_conv_kernels_match_torch_fp32(case):
    """conv_stem8_kernel (pixels straight from global memory) / conv_fewout_kernel (halo patch in LDS), filter as the MFMA's A operand, at
    the step's own sizes and at ragged ones, against torch's fp32 convolution of the same bf16-valued operands; the
    InstanceNorm partials of the stem through o2m_instnorm_finalize against the fp32 moments of that reference.  The
    launch timer names the kernel that ran."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, k, pad, reflect, extras, kernel = case
    torch.manual_seed(23)
    dev, dt = "cuda", torch.bfloat16
    x = torch.randn(B, Hh, Ww, Ci, device=dev).to(dt)
    w = (torch.randn(Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5).to(dt)
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.full((B, ho, wo, Co), float("nan"), device=dev, dtype=dt)
    bias = torch.randn(Co, device=dev) if "bias" in extras else None
    act = H.ACT_LRELU if "lrelu" in extras else H.ACT_NONE
    mode = H.PAD_REFLECT if reflect else H.PAD_ZERO
    part = rows = None
    if "stats" in extras:
        rows = H.conv2d_stats_rows(x, w, y, pad=pad)
        assert rows > 0 and (ho * wo) % rows == 0
        part = torch.full((B * (ho * wo // rows) * Co * 2,), float("nan"), device=dev)
    H.launch_timing(True)
    try:
        H.conv2d_fwd(x, w, y, bias=bias, pad=pad, pad_mode=mode, act=act, stats=part)
        names = set(H.launch_timing_read())
    finally:
        H.launch_timing(False)
    assert names == {kernel}, names
    xin = x.float().permute(0, 3, 1, 2)
    xin = F.pad(xin, (pad,) * 4, mode="reflect") if reflect else F.pad(xin, (pad,) * 4)
    ref = F.conv2d(xin, w.float().permute(0, 3, 1, 2), bias)
    if part is not None:
        mr = torch.empty(B, Co, 2, device=dev)
        H.instnorm_finalize(part, mr, ho * wo, ho * wo // rows, 1e-5)
        assert float((mr[..., 0] - ref.mean((2, 3))).abs().max()) < 1e-4
        rstd = (ref.var((2, 3), unbiased=False) + 1e-5).rsqrt()
        assert float(((mr[..., 1] - rstd) / rstd).abs().max()) < 1e-4
    if act == H.ACT_LRELU:
        ref = F.leaky_relu(ref, 0.2)
    ref = ref.permute(0, 2, 3, 1)
    assert torch.isfinite(y.float()).all()  # every output element written (the NaN prefill is gone)
    err = float((y.float() - ref).norm() / ref.norm())
    assert err < 3e-3, err
    worst = float((y.float() - ref).abs().max() / ref.abs().max())
    assert worst < 2e-2, worst


TRUNK_CONVS = [
    # B, H, W, Ci, Co, pad, extras            (conv_halo_any_kernel<4>: 4 x 4 taps, clipped 8 x 32 tiles, 64 channels per block)
    (4, 127, 127, 64, 128, 1, "bias+stats"),   # D / S trunk (builder.py:272): 126 x 126 out, statistics from clipped tiles
    (4, 63, 63, 128, 256, 1, "bias+stats"),    # 62 x 62
    (8, 31, 31, 256, 512, 1, "bias+stats"),    # 30 x 30: one tile column, eight channel blocks
    (16, 30, 30, 512, 256, 2, "plain"),        # data gradient of the last trunk conv: pad 2, 31 x 31 out
    (4, 126, 126, 128, 64, 2, "plain"),        # data gradient of the first: 127 x 127 out
    (2, 20, 45, 64, 64, 1, "bias+lrelu"),      # ragged both ways (routed there by the test hook)
]


@pytest.mark.parametrize("case", TRUNK_CONVS, ids=lambda c: "x".join(map(str, c[:6])) + "-" + c[6])
def test_clipped_halo_kernel_matches_torch_fp32(case):
    """The 4 x 4 trunk convolutions on odd-sized maps and their data gradients on the clipped halo-tile kernel against
    torch's fp32 convolution of the same bf16-valued operands; the InstanceNorm partials of the clipped tiles (4 per
    tile, o2m_conv2d_stats_chunks of them per sample) through o2m_instnorm_finalize against the reference's moments."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, pad, extras = case
    torch.manual_seed(29)
    dev, dt = "cuda", torch.bfloat16
    x = torch.randn(B, Hh, Ww, Ci, device=dev).to(dt)
    w = (torch.randn(Co, 4, 4, Ci, device=dev) / (Ci * 16) ** 0.5).to(dt)
    ho, wo = Hh + 2 * pad - 3, Ww + 2 * pad - 3
    y = torch.full((B, ho, wo, Co), float("nan"), device=dev, dtype=dt)
    bias = torch.randn(Co, device=dev) if "bias" in extras else None
    act = H.ACT_LRELU if "lrelu" in extras else H.ACT_NONE
    prev = H.debug_fill_blocks(1) if B == 2 else None
    try:
        part = chunks = None
        if "stats" in extras:
            chunks = H.conv2d_stats_chunks(x, w, y, pad=pad)
            assert chunks == 4 * ((ho + 7) // 8) * ((wo + 31) // 32), chunks
            part = torch.full((B * chunks * Co * 2,), float("nan"), device=dev)
        H.launch_timing(True)
        try:
            H.conv2d_fwd(x, w, y, bias=bias, pad=pad, pad_mode=H.PAD_ZERO, act=act, stats=part)
            names = set(H.launch_timing_read())
        finally:
            H.launch_timing(False)
    finally:
        if prev is not None:
            H.debug_fill_blocks(prev)
    assert names == {"conv_halo<bf16,4x4,8x32x64>"}, names
    ref = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (pad,) * 4), w.float().permute(0, 3, 1, 2), bias)
    if part is not None:
        mr = torch.empty(B, Co, 2, device=dev)
        H.instnorm_finalize(part, mr, ho * wo, chunks, 1e-5)
        assert float((mr[..., 0] - ref.mean((2, 3))).abs().max()) < 1e-4
        rstd = (ref.var((2, 3), unbiased=False) + 1e-5).rsqrt()
        assert float(((mr[..., 1] - rstd) / rstd).abs().max()) < 1e-4
    if act == H.ACT_LRELU:
        ref = F.leaky_relu(ref, 0.2)
    ref = ref.permute(0, 2, 3, 1)
    assert torch.isfinite(y.float()).all()  # every output element written
    err = float((y.float() - ref).norm() / ref.norm())
    assert err < 3e-3, err
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 2e-2


@pytest.mark.parametrize("fmt", [torch.float8_e4m3fn, torch.float8_e5m2], ids=["e4m3", "e5m2"])
def test_fp8_quantisation_matches_torch(fmt):
    """o2m_amax + o2m_quantize_fp8 (per-tensor scale FMT_MAX / amax, round to nearest even, OCP formats)
    against torch's own float8 conversion of the same scaled values."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(4)
    x = (torch.randn(4, 32, 32, 64, device="cuda") * 3).to(torch.bfloat16)
    deq2 = torch.empty(2, device="cuda")
    y = torch.empty(x.shape, dtype=fmt, device="cuda")
    H.quantize_fp8(x, y, deq2)
    deq, amax = deq2[0], deq2[1]
    top = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    assert float(amax) == float(x.float().abs().max())
    assert abs(float(deq) * top / float(amax) - 1) < 1e-6
    want = (x.float() * (top / amax)).clamp(-top, top).to(fmt)
    same = (y.view(torch.uint8) == want.view(torch.uint8)).float().mean()
    assert float(same) > 0.999, float(same)  # (a handful of exact ties may round the other way)
    back = y.float() * deq
    rel = float((back - x.float()).norm() / x.float().norm())
    assert rel < (0.04 if fmt == torch.float8_e4m3fn else 0.08), rel  # 3 / 2 mantissa bits


@pytest.mark.parametrize("shape", [(4, 32, 32, 64), (48, 64, 64, 256), (1, 8, 8, 8)], ids=["small", "decode-group", "tiny"])
def test_fp8_delayed_scaling_is_the_two_pass_result_one_call_late(shape):
    """o2m_quantize_fp8_delayed through _hip.quantize_fp8_site: the first call of a site is the two-pass form; the
    second call of the SAME tensor (scale from the first call's partial maxima) gives the same bytes and dequantisation
    factor; a third call with a tensor twice as large is scaled by the PREVIOUS amax (so it saturates at FMT_MAX), and
    a fourth call with it again is exact once more -- the partial maxima each call records are those of its tensor."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(6)
    x = (torch.randn(*shape, device="cuda") * 3).to(torch.bfloat16)
    fmt, top = torch.float8_e4m3fn, 448.0
    ref, ref_deq = torch.empty(x.shape, dtype=fmt, device="cuda"), torch.empty(2, device="cuda")
    H.quantize_fp8(x, ref, ref_deq)
    site = H.Fp8Site()
    for _ in range(2):  # two-pass, then delayed with the same tensor's maxima
        y, deq = torch.empty(x.shape, dtype=fmt, device="cuda"), torch.empty(2, device="cuda")
        H.quantize_fp8_site(x, y, deq, site)
        assert torch.equal(y.view(torch.uint8), ref.view(torch.uint8)) and torch.equal(deq, ref_deq)
    x2 = (x.float() * 2).to(torch.bfloat16)
    y, deq = torch.empty(x.shape, dtype=fmt, device="cuda"), torch.empty(2, device="cuda")
    H.quantize_fp8_site(x2, y, deq, site)  # scale of x: the upper half of x2's range saturates
    assert torch.equal(deq, ref_deq)
    assert float(y.float().abs().max()) == top
    want = (x2.float() * (top / ref_deq[1])).clamp(-top, top).to(fmt)
    assert float((y.view(torch.uint8) == want.view(torch.uint8)).float().mean()) > 0.999
    y, deq = torch.empty(x.shape, dtype=fmt, device="cuda"), torch.empty(2, device="cuda")
    H.quantize_fp8_site(x2, y, deq, site)  # now with x2's own maxima (recorded by the third call)
    assert float(deq[1]) == float(x2.float().abs().max())
    ref2, ref2_deq = torch.empty(x.shape, dtype=fmt, device="cuda"), torch.empty(2, device="cuda")
    H.quantize_fp8(x2, ref2, ref2_deq)
    assert torch.equal(y.view(torch.uint8), ref2.view(torch.uint8)) and torch.equal(deq, ref2_deq)


FP8_CONVS = [
    # B, H, W, Ci, Co, k, pad, reflect, x format, features
    (16, 64, 64, 256, 256, 3, 1, True, torch.float8_e4m3fn, "plain"),
    (16, 64, 64, 256, 256, 3, 1, True, torch.float8_e4m3fn, "modulated"),
    (16, 66, 66, 256, 256, 3, 2, False, torch.float8_e5m2, "plain"),      # data-gradient call: e5m2 gradients
    (8, 64, 64, 512, 512, 3, 1, True, torch.float8_e4m3fn, "epilogue"),
    (2, 16, 16, 128, 64, 3, 1, False, torch.float8_e4m3fn, "epilogue"),   # small: one ragged tile, Co < 256
]


@pytest.mark.parametrize("case", FP8_CONVS, ids=lambda c: "x".join(map(str, c[:7])) + "-" + c[9])
def test_fp8_conv_matches_torch_on_the_quantised_operands(case):
    """BASELINE config #5: the fp8 form of the phase-pipelined igemm kernel (e4m3 / e5m2 activations x e4m3
    filters on the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), fp32 accumulate, device-side dequantisation factors, bf16 out).
    Checked against torch's fp32 convolution of the SAME quantised operands, so only the summation order and
    the bf16 rounding of y differ (3e-3); the quantisation error itself is what test_hip_parity's fp8 mode bounds."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, k, pad, reflect, xfmt, feat = case
    torch.manual_seed(23)
    dev = "cuda"
    x = torch.randn(B, Hh, Ww, Ci, device=dev).to(torch.bfloat16)
    nw = B if feat == "modulated" else 1
    w = (torch.randn(nw, Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5).to(torch.bfloat16)
    dq = torch.empty(2, 2, device=dev)  # rows: {1 / scale, amax} of x and of w
    x8 = torch.empty(x.shape, dtype=xfmt, device=dev)
    w8 = torch.empty(w.shape, dtype=torch.float8_e4m3fn, device=dev)
    H.quantize_fp8(x, x8, dq[0])
    H.quantize_fp8(w, w8, dq[1])
    deq = dq.view(-1)  # {1/scale_x, amax_x, 1/scale_w, amax_w}
    ho, wo = Hh + 2 * pad - k + 1, Ww + 2 * pad - k + 1
    y = torch.empty(B, ho, wo, Co, device=dev, dtype=torch.bfloat16)
    kw = dict(pad=pad, pad_mode=H.PAD_REFLECT if reflect else H.PAD_ZERO, act=H.ACT_NONE, deq=deq)
    scale = bias = res = None
    if feat == "modulated":
        scale = torch.rand(B, Co, device=dev) + 0.5
        H.conv2d_fwd(x8, w8, y, out_scale=scale, per_sample_w=True, **kw)
    elif feat == "epilogue":
        bias = torch.randn(Co, device=dev)
        res = torch.randn(B, ho, wo, Co, device=dev).to(torch.bfloat16)
        kw["act"] = H.ACT_LRELU
        H.conv2d_fwd(x8, w8[0], y, bias=bias, residual=res, **kw)
    else:
        H.conv2d_fwd(x8, w8[0], y, **kw)
    xq, wq = x8.float() * deq[0], w8.float() * deq[2]
    xin = xq.permute(0, 3, 1, 2)
    xin = F.pad(xin, (pad,) * 4, mode="reflect") if reflect else F.pad(xin, (pad,) * 4)
    if feat == "modulated":
        ref = F.conv2d(xin.reshape(1, B * Ci, *xin.shape[2:]), wq.permute(0, 1, 4, 2, 3).reshape(B * Co, Ci, k, k),
                       groups=B).view(B, Co, ho, wo) * scale.view(B, Co, 1, 1)
    else:
        ref = F.conv2d(xin, wq[0].permute(0, 3, 1, 2))
    if feat == "epilogue":
        ref = F.leaky_relu(ref + bias.view(1, -1, 1, 1), 0.2) + res.float().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    err = float((y.float() - ref).norm() / ref.norm())
    assert err < 3e-3, err
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 2e-2
    # and the quantised conv stays within fp8's own error of the unquantised one
    full = F.conv2d(F.pad(x.float().permute(0, 3, 1, 2), (pad,) * 4, mode="reflect" if reflect else "constant"),
                    w[0].float().permute(0, 3, 1, 2)) if feat == "plain" else None
    if full is not None:
        qerr = float((y.float() - full.permute(0, 2, 3, 1)).norm() / full.norm())
        assert qerr < 8e-2, qerr


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_rgb_tail_space_to_depth_path_is_self_adjoint(precision):
    """The 64 -> 3 7x7 reflect-padded tail conv runs as a stride-4 conv over 4x4 output blocks
    (forward AND weight gradient; ops.PreparedWeight.s2d_*), its data gradient as a plain conv:
    the three must stay mutually adjoint at full size (config #4's 512x512, B = 8)."""
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.model import layers as L

    o2m.set_precision(precision)
    torch.manual_seed(5)
    conv = L.ReflectFused(L.EqualisedConv2d(64, 3, 7, padding=0), 3).to("cuda")
    x = torch.randn(8, 64, 512, 512, device="cuda", requires_grad=True)
    g = torch.randn(8, 3, 512, 512, device="cuda")
    y = conv(x)
    (y.float() * g).sum().backward()
    weight, bias = conv.conv.weight.weight, conv.conv.bias
    # the forward is linear in W (+ bias): <y - bias, g> = <x, gx> = <W, gW>
    lhs = float((y.double() * g.double()).sum() - (bias.detach().double().view(1, 3, 1, 1) * g.double()).sum())
    via_x = float((x.grad.double() * x.detach().double()).sum())
    via_w = float((weight.grad.double() * weight.detach().double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    tol = 5e-5 if precision == "fp32" else 3e-3
    assert abs(lhs - via_x) < tol * scale, (lhs, via_x, scale)
    assert abs(lhs - via_w) < tol * scale, (lhs, via_w, scale)
    assert float((bias.grad.double() - g.double().sum((0, 2, 3))).norm() / g.double().sum((0, 2, 3)).norm()) < tol


@pytest.mark.parametrize("modulated", [False, True], ids=["conv", "modconv"])
def test_layer_used_k_times_accumulates_k_weight_gradients(modulated):
    """One filter applied K times inside ONE backward pass (a decoder layer sees five batches per
    generator step): the kernel-layout accumulators + per-layer finalisation must yield exactly the
    sum of the K single-use gradients -- filter, to_style weight and bias alike."""
    import one_to_many_gan_amd as o2m
    from one_to_many_gan_amd.model import layers as L

    o2m.set_precision("fp32")
    torch.manual_seed(11)
    K, B, C, S = 3, 2, 16, 12
    layer = (L.Conv2dWeightModulate(C, C, 3, 6, 1) if modulated else L.EqualisedConv2d(C, C, 3, padding=1)).to("cuda")
    xs = [torch.randn(B, C, S, S, device="cuda") for _ in range(K)]
    ws = [torch.rand(B, 6, device="cuda") for _ in range(K)]
    gs = [torch.randn(B, C, S, S, device="cuda") for _ in range(K)]
    call = (lambda k: layer(xs[k], ws[k])) if modulated else (lambda k: layer(xs[k]))
    params = [p for p in layer.parameters()]
    singles = [torch.zeros_like(p) for p in params]
    for k in range(K):
        for p in params:
            p.grad = None
        (call(k).float() * gs[k]).sum().backward()
        for acc, p in zip(singles, params):
            acc += p.grad
    for p in params:
        p.grad = None
    sum((call(k).float() * gs[k]).sum() for k in range(K)).backward()
    for want, p, (name, _) in zip(singles, params, layer.named_parameters()):
        err = float((p.grad - want).norm() / want.norm())
        assert err < 1e-5, (name, err)


def test_c_abi_rejects_bad_arguments():
    """Error behaviour of the boundary: bad shapes are refused before any launch."""
    from one_to_many_gan_amd import _hip as H

    x = torch.zeros(1, 8, 8, 8, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(8, 3, 3, 8, device="cuda", dtype=torch.bfloat16)
    y = torch.zeros(1, 8, 8, 8, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="10001"):  # reflect pad must be < H
        H.conv2d_fwd(x, w, torch.zeros(1, 22, 22, 8, device="cuda", dtype=torch.bfloat16), pad=8,
                     pad_mode=H.PAD_REFLECT, act=0)
    with pytest.raises(RuntimeError, match="10001"):  # channels must be a multiple of 8
        H.conv2d_fwd(torch.zeros(1, 8, 8, 4, device="cuda", dtype=torch.bfloat16),
                     torch.zeros(8, 3, 3, 4, device="cuda", dtype=torch.bfloat16), y, pad=1, pad_mode=0, act=0)
    with pytest.raises(RuntimeError, match="GPU"):
        H.conv2d_fwd(x.cpu(), w, y, pad=1, pad_mode=0, act=0)
    with pytest.raises(RuntimeError, match="contiguous"):
        H.conv2d_fwd(x.permute(0, 2, 1, 3), w, y, pad=1, pad_mode=0, act=0)
    with pytest.raises(RuntimeError, match="bfloat16 or float32"):
        H.conv2d_fwd(x.half(), w.half(), y.half(), pad=1, pad_mode=0, act=0)
    with pytest.raises(RuntimeError, match="workspace too small"):
        H.instnorm_stats(y, torch.zeros(4, device="cuda"), torch.zeros(1, 8, 2, device="cuda"), 1e-5)
    ws = torch.zeros(H.instnorm_ws_floats(1, 64, 8), device="cuda")
    with pytest.raises(RuntimeError, match="10001"):  # tanh has no InstanceNorm backward here
        H.instnorm_bwd(y, y, torch.zeros(1, 8, 2, device="cuda"), ws, torch.zeros(1, 8, 2, device="cuda"), y, H.ACT_TANH)


def test_c_abi_rejects_bad_arguments_of_the_widened_entry_points():
    """o2m_gather_images / o2m_ada_* / o2m_reflect_fold / o2m_prepare_weights / o2m_resample2d refuse
    malformed calls before launching anything."""
    from one_to_many_gan_amd import _hip as H

    dev = "cuda"
    img = torch.zeros(2, 8, 8, 8, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="uint8"):  # pool must be uint8
        H.gather_images(torch.zeros(4, 8, 8, 3, device=dev), torch.zeros(2, dtype=torch.int32, device=dev),
                        torch.zeros(2, dtype=torch.uint8, device=dev), img)
    with pytest.raises(RuntimeError, match="10001"):  # colour transforms: 1 or 3 real channels
        H.ada_colour(img, torch.zeros(2, 3, 4, device=dev), torch.empty_like(img), 2)
    with pytest.raises(RuntimeError, match="10001"):  # a reflection margin must be smaller than the image
        H.reflect_fold(torch.zeros(2, 8 + 8, 8, 8, device=dev, dtype=torch.bfloat16), img, 8, 0)
    with pytest.raises(RuntimeError, match="dtype"):  # gather adjoint writes the gradient's dtype
        H.ada_grid_sample_bwd(img, torch.zeros(2, 6, device=dev), torch.zeros(2, 8, 8, 8, device=dev))
    with pytest.raises(RuntimeError, match="10002"):  # per-output kernel: square tap counts or a 1-D pass
        H.resample2d(img, torch.empty(2, 8, 8, 8, device=dev, dtype=torch.bfloat16),
                     torch.zeros(8, dtype=torch.int32, device=dev), torch.ones(8, 2, device=dev),
                     torch.zeros(8, dtype=torch.int32, device=dev), torch.ones(8, 3, device=dev), 2, 3, 0, 0)
    w = torch.zeros(8, 8, 3, 3, device=dev)
    with pytest.raises(RuntimeError, match="come together"):  # q and qt come as a pair
        H.prepare_weights(w, torch.empty(8, 3, 3, 8, device=dev), torch.empty(8, 3, 3, 8, device=dev, dtype=torch.bfloat16),
                          torch.empty(8, 3, 3, 8, device=dev, dtype=torch.bfloat16), torch.empty(8, 8, device=dev), None, 0.1)


@pytest.mark.gpu
def test_channel_sums_deterministic_mode_is_bitwise_reproducible_and_matches_atomics():
    """deterministic_cuda_kernels (reference train.py:41-45): act_bwd_reduce / fold_scale_dot add their
    per-chunk rows in chunk order instead of with fp32 atomics.  Same numbers as the atomic form to fp32
    rounding, identical bits from launch to launch; the latent gradient of style_bwd has one fixed-order
    form in both modes."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(5)
    B, S, C = 16, 64, 256
    g = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    y = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    x = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    scale = torch.rand(B, C, device="cuda") + 0.5

    def run():
        sums = torch.zeros(B, 2, C, device="cuda")
        dots = torch.zeros(B, C, device="cuda")
        gu, gx = torch.empty_like(g), torch.empty_like(g)
        H.act_bwd_reduce(g, y, None, None, gu, sums, H.ACT_LRELU)
        H.fold_scale_dot(g, x, scale, gx, dots, 0)
        return sums, dots, gu, gx

    assert not H.DETERMINISTIC
    ref = run()
    H.DETERMINISTIC = True
    try:
        a, b = run(), run()
        # accumulation semantics: a second launch ADDS to sums
        twice = a[0].clone()
        H.act_bwd_reduce(g, y, None, None, torch.empty_like(g), twice, H.ACT_LRELU)
    finally:
        H.DETERMINISTIC = False
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    for u, v in zip(a, ref):
        assert float((u.float() - v.float()).abs().max()) <= 2e-5 * float(v.float().abs().max())
    assert float((twice - 2 * a[0]).abs().max()) <= 1e-6 * float(a[0].abs().max())
    # exact values: fp64 sums of the same bf16 operands
    d = g.double() * torch.where(y > 0, 1.0, 0.2).double()
    want0 = d.sum(dim=(1, 2))
    assert float((a[0][:, 0].double() - want0).abs().max()) <= 1e-5 * float(want0.abs().max())
    want_dots = (g.double() * x.double()).sum(dim=(1, 2))
    assert float((a[1].double() - want_dots).abs().max()) <= 1e-5 * float(want_dots.abs().max())


@pytest.mark.gpu
def test_launch_timer_files_every_kernel_of_a_call_under_its_own_name():
    """bench.py's roofline object times KERNELS, not calls: the data gradient of a reflect-padded 64x64 layer
    (66x66 outputs = 273 tiles) runs as the phase-pipelined kernel over 256 tiles plus a 128x128 launch over
    the tail rows, and the two must be filed separately with the algorithmic FLOP of the rows each covers."""
    from one_to_many_gan_amd import _hip as H

    B, S, C = 16, 66, 256
    x = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(C, 3, 3, C, device="cuda") / 48).to(torch.bfloat16)
    y = torch.empty(B, S, S, C, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(B, S, S, C, device="cuda").to(torch.bfloat16)
    dw = torch.zeros(C, 3, 3, C, device="cuda")
    H.conv2d_fwd(x, w, y, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)  # untimed
    assert H.launch_timing(True) is False
    try:
        H.conv2d_fwd(x, w, y, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
        H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_ZERO)
        stats = H.launch_timing_read()
    finally:
        H.launch_timing(False)
    rows = B * S * S
    per_row = 2.0 * C * 9 * C
    p8 = stats["conv_igemm_p8<bf16,256x256>"]
    tail = stats["conv_igemm<bf16,128x128,in_scale=0>"]
    assert p8[0] == 1 and tail[0] == 1
    assert p8[2] == 256 * 256 * per_row and tail[2] == (rows - 256 * 256) * per_row
    assert 0 < tail[1] < p8[1] < 1e-3
    wg = [k for k in stats if k.startswith("conv_wgrad")]  # (whichever weight-gradient kernel the shape selects)
    assert len(wg) == 1 and stats[wg[0]][2] == rows * per_row and stats["wgrad_reduce"][0] == 1
    assert H.launch_timing_read() == {}  # reading clears the records


# ---------------------------------------------------------------------------------- round 3: fused backward forms


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_fold_scale_dot_residual_and_fused_activation_match_the_separate_kernels(dt):
    """o2m_fold_scale_dot with gres (residual gradient added) and with the fused activation backward of the layer
    below (act_sums) against the chain it replaces: fold_scale_dot -> elementwise add -> act_bwd_reduce."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(11)
    B, S, C, pad = 3, 20, 64, 1
    gpad = torch.randn(B, S + 2, S + 2, C, device="cuda").to(dt)
    x = torch.relu(torch.randn(B, S, S, C, device="cuda")).to(dt)  # the output of a ReLU layer
    s = torch.randn(B, C, device="cuda")
    dmul = torch.rand(B, C, device="cuda") + 0.5
    gres = torch.randn(B, S, S, C, device="cuda").to(dt)
    tol = 1e-6 if dt == torch.float32 else 2e-2

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    # separate kernels
    gx0, dots0 = torch.empty_like(x), torch.zeros(B, C, device="cuda")
    H.fold_scale_dot(gpad, x, s, gx0, dots0, pad)
    # (a) residual gradient
    gx1, dots1 = torch.empty_like(x), torch.zeros(B, C, device="cuda")
    H.fold_scale_dot(gpad, x, s, gx1, dots1, pad, gres=gres)
    assert rel(gx1, gx0.float() + gres.float()) < tol
    assert rel(dots1, dots0) < 1e-5
    # (b) fused activation backward: reference = act_bwd_reduce on the stored gradient
    for act in (H.ACT_RELU, H.ACT_LRELU):
        gu0, sums0 = torch.empty_like(x), torch.zeros(B, 2, C, device="cuda")
        H.act_bwd_reduce(gx0, x, None, dmul, gu0, sums0, act)
        gu1, sums1, dots2 = torch.empty_like(x), torch.zeros(B, 2, C, device="cuda"), torch.zeros(B, C, device="cuda")
        H.fold_scale_dot(gpad, x, s, gu1, dots2, pad, act=act, act_mul=dmul, act_sums=sums1)
        assert rel(gu1, gu0) < tol, act
        assert rel(sums1, sums0) < (1e-5 if dt == torch.float32 else 2e-2), act
        assert rel(dots2, dots0) < 1e-5
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 128), (4, 128, 128, 64, 256), (3, 64, 64, 128, 64),
                                   (2, 256, 256, 64, 128), (8, 128, 128, 128, 64)],
                         ids=["256x64-tile", "p8", "co64", "halo128", "halo64"])
def test_data_gradient_epilogue_emits_style_scale_and_dot(shape):
    """O2M_STATS_DOT: the data gradient of a zero-padded modulated conv leaves the GEMM epilogue multiplied by the
    style and with its style dot as row-block partials -- against conv + o2m_fold_scale_dot(pad 0)."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(12)
    B, Hh, Ww, Ck, Cn = shape  # reduction channels (the conv's outputs), result channels (the conv's inputs)
    gu = torch.randn(B, Hh, Ww, Ck, device="cuda").bfloat16()
    w_d = (torch.randn(Cn, 3, 3, Ck, device="cuda") / (3 * Ck ** 0.5)).bfloat16()
    x = torch.randn(B, Hh, Ww, Cn, device="cuda").bfloat16()
    s = torch.randn(B, Cn, device="cuda")
    gxp = torch.empty(B, Hh, Ww, Cn, device="cuda", dtype=torch.bfloat16)
    H.conv2d_fwd(gu, w_d, gxp, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
    gx0, dots0 = torch.empty_like(x), torch.zeros(B, Cn, device="cuda")
    H.fold_scale_dot(gxp, x, s, gx0, dots0, 0)
    rows = H.conv2d_stats_rows(gu, w_d, x, pad=1)
    assert rows > 0 and (Hh * Ww) % rows == 0
    nchunks = Hh * Ww // rows
    part = torch.full((B * nchunks * Cn * 2,), float("nan"), device="cuda")
    gx1, dots1, xs1 = torch.empty_like(x), torch.empty(B, Cn, device="cuda"), torch.empty_like(x)
    H.conv2d_fwd(gu, w_d, gx1, out_scale=s, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, stats=part, aux=x,
                 aux_scaled=xs1)
    H.conv2d_dots_finalize(part, dots1, nchunks)
    torch.cuda.synchronize()
    assert torch.equal(xs1, (x.float() * s[:, None, None, :]).bfloat16()), "aux_scaled = the modulated input x * s"
    # gx0 went through a bf16 rounding of the unscaled gradient first: one bf16 ulp of slack
    assert float((gx1.double() - gx0.double()).norm() / gx0.double().norm()) < 6e-3
    # the fused dot uses the fp32 accumulator, the separate one the rounded gradient
    assert float((dots1.double() - dots0.double()).norm() / dots0.double().norm()) < 5e-3
    # and against fp64 of the same operands
    ref = torch.nn.functional.conv2d(gu.double().cpu().permute(0, 3, 1, 2), w_d.double().cpu().permute(0, 3, 1, 2),
                                     padding=1).permute(0, 2, 3, 1)
    dref = (ref * x.double().cpu()).sum(dim=(1, 2))
    assert float((dots1.double().cpu() - dref).norm() / dref.norm()) < 1e-4
    assert float((gx1.double().cpu() - ref * s.double().cpu()[:, None, None, :]).norm() / ref.norm()) < 6e-3


def test_wgrad_in_scale_equals_the_prescaled_input():
    """The weight-gradient kernel scaling x by the style while staging it (in_scale) against the stored x * s."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(13)
    for (B, S, Ci, Co, pad_mode) in ((3, 32, 64, 64, H.PAD_REFLECT), (2, 64, 128, 256, H.PAD_ZERO), (2, 33, 64, 128, H.PAD_ZERO)):
        x = torch.randn(B, S, S, Ci, device="cuda").bfloat16()
        g = torch.randn(B, S, S, Co, device="cuda").bfloat16()
        s = torch.randn(B, Ci, device="cuda")
        xs = (x.float() * s[:, None, None, :]).bfloat16()
        a = torch.zeros(Co, 3, 3, Ci, device="cuda")
        b = torch.zeros_like(a)
        H.conv2d_wgrad(xs, g, a, pad=1, pad_mode=pad_mode)
        H.conv2d_wgrad(x, g, b, in_scale=s, pad=1, pad_mode=pad_mode)
        torch.cuda.synchronize()
        assert float((a - b).norm() / a.norm()) < 1e-6, (B, S, Ci, Co)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_transposed_upsample_tile_kernel_matches_the_dense_operator(dt):
    """The one-launch LDS tile kernel of the 6-tap transposed upsample against the dense operator matrices."""
    from one_to_many_gan_amd import _hip as H
    from one_to_many_gan_amd import resample as R

    torch.manual_seed(14)
    for (B, n_h, n_w, C) in ((2, 16, 24, 32), (1, 64, 64, 128), (2, 9, 13, 8), (1, 32, 32, 256)):
        sy, wy, sx, wx, T, ho, wo = R.taps("up", n_h, n_w, True, torch.device("cuda"))
        assert int(T) == 6 and (ho, wo) == (n_h, n_w)
        g = torch.randn(B, 2 * n_h, 2 * n_w, C, device="cuda").to(dt)
        out = torch.empty(B, ho, wo, C, device="cuda", dtype=dt)
        H.resample2d(g, out, sy, wy, sx, wx, int(T), int(T), T.span_y, T.span_x)
        torch.cuda.synchronize()
        a_h = torch.from_numpy(R.operator_matrix("up", n_h).T.copy())  # [n_h][2 n_h]
        a_w = torch.from_numpy(R.operator_matrix("up", n_w).T.copy())
        ref = torch.einsum("ip,bpqc,jq->bijc", a_h, g.double().cpu(), a_w)
        err = float((out.double().cpu() - ref).norm() / ref.norm())
        assert err < (1e-6 if dt == torch.float32 else 4e-3), (B, n_h, n_w, C, err)


@pytest.mark.parametrize("co", [64, 128])
def test_halo_kernel_instance_norm_partials_match_the_statistics_pass(co):
    """The halo-tile kernel's InstanceNorm partials (one per wave row of an 8 x 32 tile: NOT consecutive pixels)
    through o2m_instnorm_finalize against the separate statistics pass over y."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(15)
    B, S, Ci = 2, 256, 64
    x = torch.randn(B, S, S, Ci, device="cuda").bfloat16()
    w = (torch.randn(co, 3, 3, Ci, device="cuda") / (3 * Ci ** 0.5)).bfloat16()
    bias = torch.randn(co, device="cuda")
    y = torch.empty(B, S, S, co, device="cuda", dtype=torch.bfloat16)
    rows = H.conv2d_stats_rows(x, w, y, pad=1)
    assert rows == 64
    nchunks = S * S // rows
    part = torch.full((B * nchunks * co * 2,), float("nan"), device="cuda")
    H.conv2d_fwd(x, w, y, bias=bias, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, stats=part)
    mr = torch.empty(B, co, 2, device="cuda")
    H.instnorm_finalize(part, mr, S * S, nchunks, 1e-5)
    ws = torch.empty(H.instnorm_ws_floats(B, S * S, co), device="cuda")
    mr0 = torch.empty(B, co, 2, device="cuda")
    H.instnorm_stats(y, ws, mr0, 1e-5)
    torch.cuda.synchronize()
    assert torch.isfinite(mr).all()
    # (the epilogue sums the fp32 accumulators, the pass the bf16-rounded y)
    assert float((mr[..., 0] - mr0[..., 0]).abs().max()) < 2e-3
    assert float(((mr[..., 1] - mr0[..., 1]) / mr0[..., 1]).abs().max()) < 2e-3


@pytest.mark.parametrize("reflect", [True, False], ids=["reflect", "zero"])
def test_p8_kernel_instance_norm_partials_match_the_statistics_pass(reflect):
    """The phase-pipelined 256 x 256 kernel's InstanceNorm partials at the shape the step selects it at (256 -> 256,
    64 x 64, B = 32 = two rounds of 256 tiles): the accumulator-direct epilogue sums each wave's 128 pixels x 64 permuted
    channels with DPP row rotations (conv_igemm.hip); through o2m_instnorm_finalize against the separate statistics pass
    over y, and y itself against torch on the same bf16 operands.  The launch timer names the kernel that ran."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(16)
    B, S, C = 32, 64, 256
    x = torch.randn(B, S, S, C, device="cuda").bfloat16()
    w = (torch.randn(C, 3, 3, C, device="cuda") / (3 * C ** 0.5)).bfloat16()
    bias = torch.randn(C, device="cuda")
    y = torch.empty(B, S, S, C, device="cuda", dtype=torch.bfloat16)
    mode = H.PAD_REFLECT if reflect else H.PAD_ZERO
    rows = H.conv2d_stats_rows(x, w, y, pad=1)
    assert rows == 128
    nchunks = S * S // rows
    part = torch.full((B * nchunks * C * 2,), float("nan"), device="cuda")
    H.launch_timing(True)
    try:
        H.conv2d_fwd(x, w, y, bias=bias, pad=1, pad_mode=mode, act=H.ACT_NONE, stats=part)
        names = set(H.launch_timing_read())
    finally:
        H.launch_timing(False)
    assert names == {"conv_igemm_p8<bf16,256x256>"}, names
    mr = torch.empty(B, C, 2, device="cuda")
    H.instnorm_finalize(part, mr, S * S, nchunks, 1e-5)
    ws = torch.empty(H.instnorm_ws_floats(B, S * S, C), device="cuda")
    mr0 = torch.empty(B, C, 2, device="cuda")
    H.instnorm_stats(y, ws, mr0, 1e-5)
    torch.cuda.synchronize()
    assert torch.isfinite(mr).all()
    assert float((mr[..., 0] - mr0[..., 0]).abs().max()) < 2e-3
    assert float(((mr[..., 1] - mr0[..., 1]) / mr0[..., 1]).abs().max()) < 2e-3
    # the map itself and the statistics against torch fp32 on the same operands (independent of both kernels)
    xn = x[:4].float().permute(0, 3, 1, 2)
    xn = F.pad(xn, (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xn, w.float().permute(0, 3, 1, 2), bias)
    got = y[:4].float().permute(0, 3, 1, 2)
    assert float((got - ref).norm() / ref.norm()) < 4e-3
    assert float((mr[:4, :, 0] - ref.mean((2, 3))).abs().max()) < 2e-3
    rstd = (ref.var((2, 3), unbiased=False) + 1e-5).rsqrt()
    assert float(((mr[:4, :, 1] - rstd) / rstd).abs().max()) < 2e-3


@pytest.mark.parametrize("reflect", [True, False], ids=["reflect", "zero"])
def test_phase_pipelined_weight_gradient_matches_torch(reflect):
    """conv_wgrad_p8_kernel (256 -> 256, 3 x 3, 64-pixel rows; transposing LDS reads, LDS-DMA fills, slabs) against
    torch's fp32 weight gradient of the same bf16-valued operands, for both paddings and an uneven last slice."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(21)
    for B in (4, 7):  # 256 / 448 image rows over 28 slices: even / uneven
        x = torch.randn(B, 64, 64, 256, device="cuda").bfloat16()
        g = torch.randn(B, 64, 64, 256, device="cuda").bfloat16()
        dw = torch.zeros(256, 3, 3, 256, device="cuda")
        pm = H.PAD_REFLECT if reflect else H.PAD_ZERO
        H.launch_timing(True)
        try:
            H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=pm, p8=True)
            torch.cuda.synchronize()
            names = set(H.launch_timing_read())
        finally:
            H.launch_timing(False)
        assert "conv_wgrad_p8<bf16,256x256>" in names, names
        xin = x.float().permute(0, 3, 1, 2)
        xin = F.pad(xin, (1,) * 4, mode="reflect") if reflect else F.pad(xin, (1,) * 4)
        ref = torch.nn.grad.conv2d_weight(xin, (256, 256, 3, 3), g.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        err = float((dw - ref).norm() / ref.norm())
        assert err < 2e-5, (B, reflect, err)
        # every tap and channel block individually (a mis-staged region would hide in the norm of the whole)
        per_tap = ((dw - ref) ** 2).sum(dim=(0, 3)).sqrt() / (ref ** 2).sum(dim=(0, 3)).sqrt()
        assert float(per_tap.max()) < 5e-5, per_tap


@pytest.mark.parametrize("shape", [(16, 64, 64, 256, 256), (3, 20, 70, 64, 128), (2, 8, 4, 32, 64)], ids=str)
def test_reflect_border_completes_the_cropped_data_gradient(shape):
    """o2m_conv2d_reflect_border: the zero-padded data-gradient conv on the cropped domain + the border ring = the
    adjoint of conv(ReflectionPad2d(1)(x)), i.e. torch's own gradient of that composition in fp32 on the same
    bf16-valued operands -- at the step's shape (256 -> 256 at 64 x 64: the phase-pipelined kernel with exactly 256 tiles),
    on a wide ragged map and on the smallest one."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Cg, Cx = shape  # Cg: channels of the incoming gradient (the conv's outputs), Cx: of the conv's input
    torch.manual_seed(41)
    g = torch.randn(B, Hh, Ww, Cg, device="cuda").bfloat16()
    w = (torch.randn(Cg, Cx, 3, 3, device="cuda") / (9 * Cx) ** 0.5).bfloat16()  # the layer's filter [Co][Ci][3][3]
    # data-gradient filter: [Ci][ky][kx][Co] flipped (ops.PreparedWeight w_d)
    w_d = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()
    gx = torch.empty(B, Hh, Ww, Cx, device="cuda", dtype=torch.bfloat16)
    H.launch_timing(True)
    try:
        H.conv2d_fwd(g, w_d, gx, pad=1, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
        H.conv2d_reflect_border(g, w_d, gx)
        names = set(H.launch_timing_read())
    finally:
        H.launch_timing(False)
    assert "conv_reflect_border<bf16,3x3>" in names, names
    if shape[0] == 16:
        assert names == {"conv_igemm_p8<bf16,256x256>", "conv_reflect_border<bf16,3x3>"}, names  # one round, no tail launch
    x = torch.zeros(B, Cx, Hh, Ww, device="cuda", requires_grad=True)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w.float())
    (y * g.float().permute(0, 3, 1, 2)).sum().backward()
    ref = x.grad.permute(0, 2, 3, 1)
    err = float((gx.float() - ref).norm() / ref.norm())
    assert err < 4e-3, err
    ring = torch.zeros(Hh, Ww, dtype=torch.bool, device="cuda")
    ring[1], ring[-2], ring[:, 1], ring[:, -2] = True, True, True, True
    err_ring = float((gx.float() - ref)[:, ring].norm() / ref[:, ring].norm())
    assert err_ring < 8e-3, err_ring  # (targets of the ring: one rounding per atomic add)


HALO_WGRADS = [
    # B, H, W, Ci, Co, k                   (conv_wgrad_halo_kernel: 3 x 3 / 4 x 4, zero pad 1, 64-channel multiples)
    (2, 256, 256, 128, 64, 3),    # decoder 128 -> 64 at 256 x 256: two filter blocks, 8 tiles per slice
    (3, 128, 128, 256, 128, 3),   # decoder 256 -> 128 at 128 x 128: eight filter blocks, an uneven last slice
    (2, 128, 128, 128, 256, 3),   # encoder 128 -> 256
    (5, 64, 96, 64, 64, 3),       # one filter block, image borders in most tiles, 6 tiles per sample
    (4, 50, 70, 64, 64, 3),       # tiles clipped on both axes
    (4, 127, 127, 64, 128, 4),    # D / S trunk (builder.py:272): 4 x 4 taps, 126 x 126 gradient map
    (4, 63, 63, 128, 256, 4),     # 62 x 62
    (8, 31, 31, 256, 512, 4),     # 30 x 30: one clipped tile column, 32 filter blocks
]


@pytest.mark.parametrize("case", HALO_WGRADS, ids=lambda c: "x".join(map(str, c)))
def test_halo_tile_weight_gradient_matches_torch(case):
    """conv_wgrad_halo_kernel (all taps of a 64 x 64 filter block per workgroup; G tile and input patch resident in
    LDS, inline-asm transposing reads, double-buffered tiles, slabs) against torch's fp32 weight gradient of the same
    bf16-valued operands; every tap and 64-channel block individually; bitwise repeatable."""
    import torch.nn.functional as F

    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Ci, Co, k = case
    kernel = f"conv_wgrad_halo<bf16,64x{k * k}x64>"
    ho, wo = Hh + 2 - k + 1, Ww + 2 - k + 1
    torch.manual_seed(31)
    x = torch.randn(B, Hh, Ww, Ci, device="cuda").bfloat16()
    g = torch.randn(B, ho, wo, Co, device="cuda").bfloat16()
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    H.launch_timing(True)
    try:
        H.conv2d_wgrad(x, g, dw, pad=1, pad_mode=H.PAD_ZERO)
        torch.cuda.synchronize()
        names = set(H.launch_timing_read())
    finally:
        H.launch_timing(False)
    assert kernel in names, names
    xin = F.pad(x.float().permute(0, 3, 1, 2), (1,) * 4)
    ref = torch.nn.grad.conv2d_weight(xin, (Co, Ci, k, k), g.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    err = float((dw - ref).norm() / ref.norm())
    assert err < 2e-5, err
    blocks = (dw - ref).view(Co // 64, 64, k * k, Ci // 64, 64)
    refb = ref.reshape(Co // 64, 64, k * k, Ci // 64, 64)
    per_block = (blocks ** 2).sum(dim=(1, 4)).sqrt() / (refb ** 2).sum(dim=(1, 4)).sqrt()
    assert float(per_block.max()) < 5e-5, per_block
    again = torch.zeros_like(dw)
    H.conv2d_wgrad(x, g, again, pad=1, pad_mode=H.PAD_ZERO)
    assert torch.equal(again, dw)
    # accumulates into dw like the other forms (a layer used twice in one backward)
    H.conv2d_wgrad(x, g, again, pad=1, pad_mode=H.PAD_ZERO)
    assert float((again - 2 * ref).norm() / ref.norm()) < 4e-5
    if k != 3:
        return
    # in_scale (the modulated convs): the style folded into the per-sample partials, x * s never materialised
    sc = torch.rand(B, Ci, device="cuda") * 2 - 0.5
    dws = torch.zeros_like(dw)
    H.launch_timing(True)
    try:
        H.conv2d_wgrad(x, g, dws, in_scale=sc, pad=1, pad_mode=H.PAD_ZERO)
        torch.cuda.synchronize()
        names = set(H.launch_timing_read())
    finally:
        H.launch_timing(False)
    assert kernel in names, names
    xs = F.pad((x.float() * sc.view(B, 1, 1, Ci)).permute(0, 3, 1, 2), (1,) * 4)
    refs = torch.nn.grad.conv2d_weight(xs, (Co, Ci, 3, 3), g.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert float((dws - refs).norm() / refs.norm()) < 2e-5


def test_tensors_beyond_2_gib_run_as_batch_slices():
    """fp32 parity mode at config #4 sizes: an input of exactly 2 GiB (16 x 512 x 512 x 128 floats) used to be rejected
    (buffer descriptors address < 2 GiB).  The launchers now cut the batch into slices; a smaller-shaped stand-in with
    the same byte count (fp32, 16 x 256 x 256 x 512 is 2 GiB too) against the same call made per 8-sample half."""
    from one_to_many_gan_amd import _hip as H

    torch.manual_seed(3)
    B, S, Ci, Co = 16, 256, 512, 8
    x = torch.randn(B, S, S, Ci, device="cuda")  # 2 GiB
    assert x.numel() * 4 > 0x7fffffff
    w = torch.randn(Co, 1, 1, Ci, device="cuda") / Ci ** 0.5
    y = torch.empty(B, S, S, Co, device="cuda")
    H.conv2d_fwd(x, w, y, pad=0, pad_mode=H.PAD_ZERO, act=H.ACT_RELU)
    halves = torch.empty_like(y)
    for b0 in (0, 8):
        H.conv2d_fwd(x[b0:b0 + 8], w, halves[b0:b0 + 8], pad=0, pad_mode=H.PAD_ZERO, act=H.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(y, halves)
    g = torch.randn(B, S, S, Co, device="cuda")
    dw, dw2 = torch.zeros(Co, 1, 1, Ci, device="cuda"), torch.zeros(Co, 1, 1, Ci, device="cuda")
    H.conv2d_wgrad(x, g, dw, pad=0, pad_mode=H.PAD_ZERO)
    for b0 in (0, 8):
        H.conv2d_wgrad(x[b0:b0 + 8], g[b0:b0 + 8], dw2, pad=0, pad_mode=H.PAD_ZERO)
    torch.cuda.synchronize()
    # (the launcher cuts 15 + 1 samples, the loop 8 + 8: the pixel slices differ, so the sums agree to rounding)
    assert float((dw - dw2).norm() / dw2.norm()) < 1e-5


def test_batched_weight_preparation_equals_the_per_layer_launches():
    """ops.prepare_network: every filter of a network in ONE launch (o2m_prepare_weights_batched) against the
    per-layer o2m_prepare_weights launches, bit for bit, plain and modulated layers, padded channel counts."""
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd import ops
    from one_to_many_gan_amd.model import builder

    for precision in ("bf16", "fp32"):
        pk.set_precision(precision)
        torch.manual_seed(4)
        net = builder.Generator(3, 6, (32, 32), 8, 3, start_filters=8).cuda()
        preps = [m._prepared() for m in net.modules() if hasattr(m, "_prepared")]
        assert len(preps) >= 8
        want = [[None if t is None else t.clone() for t in p.get()] for p in preps]  # per-layer path
        ops.bump_weights_epoch([p.weight for p in preps])
        for p in preps:  # poison the buffers: the batched launch must rewrite every element
            for t in p.buffers():
                if t is not None:
                    t.fill_(float("nan"))
        ops.prepare_network(net)
        torch.cuda.synchronize()
        for p, w in zip(preps, want):
            assert p._key == p.version_key(), "marked fresh"
            for a, b in zip(p.get(), w):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b)
    pk.set_precision("bf16")


def test_batched_weight_gradient_finalisation_equals_the_per_layer_launches():
    """ops._finalize_batched: every pending layer's accumulators -> .grad in ONE launch (o2m_wgrad_finalize_batched)
    against the per-layer o2m_wgrad_finalize launches, bit for bit: plain and modulated layers (dL/dQ term), padded
    channel counts, += into an existing gradient, accumulators and dL/dQ tables cleared."""
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd import ops
    from one_to_many_gan_amd.model import builder

    pk.set_precision("bf16")
    torch.manual_seed(5)
    net = builder.Generator(3, 6, (32, 32), 8, 3, start_filters=8).cuda()
    preps = [m._prepared() for m in net.modules() if hasattr(m, "_prepared")]
    assert any(p.need_q for p in preps) and not all(p.need_q for p in preps)
    accs, gqs, g0 = [], [], []
    for p in preps:
        p.get()
        p.dw_acc = torch.zeros((p.cop, p.kh, p.kw, p.cip), device="cuda")  # (what PreparedWeights.accumulators allocates)
        p.gq_acc = torch.zeros((p.cop, p.cip), device="cuda") if p.need_q else None
        accs.append(torch.randn_like(p.dw_acc))
        gqs.append(None if p.gq_acc is None else torch.randn_like(p.gq_acc))
        g0.append(torch.randn_like(p.weight))

    def load():
        for p, a, q, g in zip(preps, accs, gqs, g0):
            p.dw_acc.copy_(a)
            if q is not None:
                p.gq_acc.copy_(q)
            p.weight.grad = g.clone()
            p.pending, p.fwd_uses, p.bwd_uses = True, 1, 1

    load()
    for p in preps:
        ops._finalize_layer(p)
    want = [p.weight.grad.clone() for p in preps]
    load()
    assert ops._finalize_batched(preps)
    torch.cuda.synchronize()
    for p, w in zip(preps, want):
        assert torch.equal(p.weight.grad, w)
        assert not p.pending and p.fwd_uses == 0 and p.bwd_uses == 0
        assert float(p.dw_acc.abs().max()) == 0.0
        assert p.gq_acc is None or float(p.gq_acc.abs().max()) == 0.0
    # a second pass through the cached job table
    load()
    assert ops._finalize_batched(preps)
    for p, w in zip(preps, want):
        assert torch.equal(p.weight.grad, w)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_fused_instance_norm_activation_downsample_matches_the_two_passes(dt):
    """o2m_instnorm_act_resample2d / o2m_instnorm_resample_bwd (InstanceNorm + (Leaky)ReLU + DownSample as one pass,
    and its backward gathering the fine gradient) against instance_norm_act followed by resample, forward and
    input gradient, on even and odd map sizes (the encoder's and the discriminator's)."""
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd import _hip as H
    from one_to_many_gan_amd import ops

    pk.set_precision("fp32" if dt == torch.float32 else "bf16")
    try:
        torch.manual_seed(31)
        # (C = 128 / 256 on even maps: the tile form of the forward -- 8 x 7 output tiles, so 36 x 20 and 64 x 64 outputs
        #  cover partial tiles in both directions and the clamped first / last tile rows)
        for (B, Hh, Ww, C, act) in ((2, 64, 64, 32, H.ACT_RELU), (3, 126, 126, 16, H.ACT_LRELU), (2, 62, 30, 64, H.ACT_LRELU),
                                    (2, 72, 40, 128, H.ACT_RELU), (2, 128, 128, 256, H.ACT_LRELU)):
            x = (torch.randn(B, Hh, Ww, C, device="cuda") * 2 + 0.3).to(dt)
            ws = torch.empty(H.instnorm_ws_floats(B, Hh * Ww, C), device="cuda")
            mr = torch.empty(B, C, 2, device="cuda")
            H.instnorm_stats(x, ws, mr, 1e-5)
            assert ops.norm_down_fusable(x, "down", mr)
            xa = x.clone().requires_grad_(True)
            xb = x.clone().requires_grad_(True)
            ya = ops.resample(ops.instance_norm_act(xa, act, eps=1e-5, stats=mr), "down")
            yb = ops.instance_norm_act_down(xb, act, "down", mr)
            g = torch.randn_like(ya)
            ya.backward(g)
            yb.backward(g)
            torch.cuda.synchronize()
            tol = 2e-6 if dt == torch.float32 else 1.5e-2  # (bf16: the two-pass form rounds the normalised map first)
            assert float((ya.float() - yb.float()).norm() / ya.float().norm()) < tol
            assert float((xa.grad.float() - xb.grad.float()).norm() / xa.grad.float().norm()) < (2e-5 if dt == torch.float32 else 3e-2)
    finally:
        pk.set_precision("bf16")


@pytest.mark.parametrize("case", [
    (16, 64, 64, 256, 256, 3, 1, True),    # p8 kernel (256 tiles) + 128x128 tail tiles, with the residual gradient
    (32, 64, 64, 256, 256, 3, 1, False),   # two rounds of p8 tiles + tail
    (3, 24, 40, 64, 128, 3, 1, True),      # generic tiles, non-square map
    (4, 40, 48, 8, 64, 7, 3, True),        # the image head's 7 x 7 (Ci = 8: the narrow-reduction loader), pad 3
], ids=["p8+tail", "p8x2", "generic", "7x7pad3"])
def test_reflect_fold_in_the_conv_epilogue_matches_the_fold_pass(case):
    """o2m_conv_desc.fold_pad: the data gradient of a conv behind ReflectionPad2d(f), folded by the GEMM's epilogue
    (plain stores + packed bf16 atomic adds on the mirrored rows / columns) against the two-pass form it replaces
    (padded GEMM output + o2m_fold_scale_dot, itself checked against torch above).  Both round to bf16 -- the fold
    pass once from an fp32 sum of bf16-rounded terms, the epilogue after every atomic add -- so pixels with several
    contributions may differ by a couple of bf16 steps; everything else must agree exactly (to one rounding of the
    residual sum where a residual is added)."""
    from one_to_many_gan_amd import _hip as H

    B, Hh, Ww, Cg, Cx, k, f, with_res = case
    torch.manual_seed(5)
    dev, dt = "cuda", torch.bfloat16
    g = torch.randn(B, Hh, Ww, Cg, device=dev).to(dt)                 # gradient of the conv output
    w = (torch.randn(Cx, k, k, Cg, device=dev) / (Cg * k * k) ** 0.5).to(dt)   # flipped / transposed filter
    res = torch.randn(B, Hh, Ww, Cx, device=dev).to(dt) if with_res else None
    kpad = k - 1
    hp, wp = Hh + 2 * f, Ww + 2 * f
    gpad = torch.empty(B, hp, wp, Cx, device=dev, dtype=dt)
    H.conv2d_fwd(g, w, gpad, pad=kpad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE)
    want = torch.empty(B, Hh, Ww, Cx, device=dev, dtype=dt)
    H.fold_scale_dot(gpad, None, None, want, None, f, gres=res)
    got = torch.full((B, Hh, Ww, Cx), float("nan"), device=dev, dtype=dt)
    H.conv2d_fwd(g, w, got, pad=kpad, pad_mode=H.PAD_ZERO, act=H.ACT_NONE, residual=res, fold_pad=f)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    single = torch.ones(Hh, Ww, dtype=torch.bool, device=dev)   # pixels with exactly one contribution
    for lo, hi in ((1, f), ):
        single[lo:hi + 1, :] = False; single[Hh - 1 - f:Hh - 1, :] = False
        single[:, lo:hi + 1] = False; single[:, Ww - 1 - f:Ww - 1] = False
    if with_res:  # (the fold pass adds the residual to the bf16-ROUNDED padded gradient, the epilogue to the fp32 sum)
        a1, b1 = got[:, single].float(), want[:, single].float()
        assert float((a1 - b1).abs().max()) <= 2.0 ** -7 * float(b1.abs().max())
    else:
        assert torch.equal(got[:, single], want[:, single])
    a, b = got[:, ~single].float(), want[:, ~single].float()
    assert float((a - b).abs().max()) <= 2.0 ** -6 * float(b.abs().max())
    assert float((a - b).norm() / b.norm()) < 4e-3


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_path_loss_tap_matches_the_two_consumer_form(dt):
    """ops.halves_sq_tap (o2m_pair_grad): a feature map taken THROUGH the path-loss pair term on its way to the next
    layer gets, in one backward pass, the gradient that autograd forms from the separate term (halves_sq_sum: reduce_bwd
    + negation) plus the next layer's gradient (accumulation add).  Same value of the term, same gradient to bf16
    rounding (the fused form rounds once instead of three times)."""
    from one_to_many_gan_amd import ops

    torch.manual_seed(11)
    dev = "cuda"
    t0 = torch.randn(8, 12, 20, 64, device=dev).to(dt)
    k = torch.randn(8, 12, 20, 64, device=dev).to(dt)   # stands for the next layer: its gradient is k
    w = torch.rand(4, device=dev) + 0.5
    a = t0.clone().requires_grad_(True)
    term_a = ops.halves_sq_sum(a, w)
    (0.37 * term_a + (a.float() * k.float()).sum()).backward()
    b = t0.clone().requires_grad_(True)
    tb, term_b = ops.halves_sq_tap(b, w)
    (0.37 * term_b + (tb.float() * k.float()).sum()).backward()
    assert torch.equal(term_a, term_b)
    ref = k.double() + 0.37 * 2 * torch.cat([(t0[:4].double() - t0[4:].double()) * w.double().view(4, 1, 1, 1),
                                            -(t0[:4].double() - t0[4:].double()) * w.double().view(4, 1, 1, 1)], 0)
    tol = 1e-2 if dt == torch.bfloat16 else 1e-5
    assert float((b.grad.double() - ref).abs().max()) <= tol * float(ref.abs().max())
    assert float((a.grad.double() - ref).abs().max()) <= 3 * tol * float(ref.abs().max())
    # the term alone (no other consumer of the map) and the map alone (term unused)
    c = t0.clone().requires_grad_(True)
    _, term_c = ops.halves_sq_tap(c, w)
    term_c.backward()
    assert float((c.grad.double() - (ref - k.double()) / 0.37).abs().max()) <= tol * float(ref.abs().max())
    e = t0.clone().requires_grad_(True)
    te, _ = ops.halves_sq_tap(e, w)
    (te.float() * k.float()).sum().backward()
    assert torch.equal(e.grad, k)
