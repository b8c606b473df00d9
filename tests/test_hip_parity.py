"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI of
libo2m_hip.so, against (a) the live CPU oracle on the same closed-form inputs and (b) the
committed fixtures produced by the reference itself.

Tolerances (relative L2 over each tensor) are stated in ``_tolerance`` below:
* precision "fp32" (fp32 storage, bf16x3 split MFMA, fp32 accumulate): the north-star gate,
  outputs within 1e-3 of the CPU reference;
* precision "bf16" (bf16 storage + bf16 MFMA, BASELINE config #2's dtype): within 3x of the
  reference's OWN bf16-autocast error on the same tensor.
"""

import json
import os

import numpy as np
import pytest
import torch

from tests.cases import CASES, host_threads, run_case
from tests.namespaces import oracle_ns, product_ns

pytestmark = pytest.mark.gpu

HOST_ONLY = {"adap", "imagebuffer", "mapping"}
NET_CASES = ("gen", "disc", "style")
_oracle_cache = {}
_YARD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bf16_yardstick.json")))
_YARD8 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fp8_yardstick.json")))
FP8_CASES = sorted(_YARD8)  # the cases with layers of >= 128 channels (tools/make_fp8_yardstick.py)


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _is_grad(key):
    return key.startswith("g")


def _tolerance(name, key, precision):
    """Relative-L2 tolerance for one tensor of one case.

    fp32 mode (bf16x3 split MFMA, ~4e-6 per op): outputs 1e-3 -- the north-star gate;
    op-level gradients 1e-3; NET-level gradients 3e-2: a forward error eps flips a fraction
    ~eps of the ReLU masks and each flip is an O(1) change of that element's gradient, so
    gradient error grows like sqrt(eps) per layer (the reference's own TF32 GPU path is far
    looser).  bf16 mode: three times the REFERENCE'S OWN bf16-autocast error on the same tensor
    (the yardstick is itself one realisation of bf16 rounding noise)
    (tests/golden/bf16_yardstick.json, tools/make_bf16_yardstick.py), floored at 1e-2 / 2e-2.
    """
    if name in HOST_ONLY:
        return 1e-4 if name == "mapping" else 1e-6
    grad = _is_grad(key)
    if precision == "fp8":
        # BASELINE config #5: three times what the same per-tensor e4m3 / e5m2 quantisation costs the
        # REFERENCE on this tensor (tests/golden/fp8_yardstick.json), floored at the single-layer level:
        # 3 mantissa bits are ~3.5e-2 on a conv output, 2 bits ~6e-2 on its data gradient -- and the
        # style / demodulation gradients, which the yardstick's reference takes from exact weight
        # gradients, come out of the fp8 data gradient here
        return max(1.5e-1 if grad else 6e-2, 3.0 * _YARD8.get(name, {}).get(key, 0.0), 3.0 * _YARD.get(name, {}).get(key, 0.0))
    if precision == "fp32":
        if grad and name.startswith(NET_CASES):
            return 3e-2
        if grad and name.endswith("_c256_b16"):
            # 16 M ReLU inputs between the two convs of the block: the handful whose pre-activation lies within the
            # forward error (~1e-5) of zero flip their mask, an O(1) change of that element's gradient each -- measured
            # 1.6e-3 on the first conv's filter gradient.  With the masks shared the same gradients are held to 1e-3
            # against the fp64 oracle (test_net_gradients_with_shared_activation_masks).
            return 5e-3
        return 1e-3
    floor = 2e-2 if grad else 1e-2
    if grad and key.endswith("bias") and name.startswith(NET_CASES):
        # a conv bias gradient is a SIGNED sum over every pixel of bf16-rounded terms: with few
        # channels (the 1- or 3-channel image convs) it cancels to a small number whose
        # relative error the per-tensor yardstick does not bound
        floor = 1e-1
    return max(floor, 3.0 * _YARD.get(name, {}).get(key, 0.0))


def _oracle(name):
    if name not in _oracle_cache:
        torch.set_num_threads(host_threads())
        _oracle_cache[name] = run_case(name, oracle_ns(), "cpu")
    return _oracle_cache[name]


def _check(name, got, want, precision, label):
    assert set(got) == set(want), (label, set(got) ^ set(want))
    bad = []
    for k, w in want.items():
        w = torch.as_tensor(w)
        assert got[k].shape == w.shape, (label, k)
        if k.endswith("/sum") or k.endswith("/sqsum"):
            continue  # signed sums cancel; the strided subsample carries the comparison
        if w.abs().max() == 0:
            if got[k].abs().max() > 1e-5:
                bad.append((k, "nonzero"))
            continue
        err, tol = _rel(got[k], w), _tolerance(name, k, precision)
        if err > tol:
            bad.append((k, err, tol))
    assert not bad, (label, bad)


STEP_CASES = [n for n in CASES if n.startswith("steps")]
OP_CASES = [n for n in CASES if n not in STEP_CASES]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", OP_CASES)
def test_hip_matches_oracle_and_fixture(name, precision, golden_dir):
    got = run_case(name, product_ns(precision), "cuda")
    _check(name, got, _oracle(name), precision, "oracle")
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    _check(name, got, {k: gold[k] for k in gold.files}, precision, "reference fixture")


@pytest.mark.parametrize("name", FP8_CASES)
def test_fp8_mode_matches_oracle_and_fixture(name, golden_dir):
    """BASELINE config #5 (fp8 weight / activation path): forward and data-gradient products of the
    >= 128-channel layers on the fp8 MFMA, everything else as in bf16 mode."""
    got = run_case(name, product_ns("fp8"), "cuda")
    _check(name, got, _oracle(name), "fp8", "oracle")
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    _check(name, got, {k: gold[k] for k in gold.files}, "fp8", "reference fixture")


def _timed_case(name, precision):
    """run_case with the library's launch timer on: (outputs, names of the MFMA conv kernels that ran)."""
    from one_to_many_gan_amd import _hip as H

    H.launch_timing(True)
    try:
        got = run_case(name, product_ns(precision), "cuda")
        names = set(H.launch_timing_read(256))
    finally:
        H.launch_timing(False)
    return got, names


@pytest.mark.parametrize("name", ["resblock_c256_b16", "modresblock_c256_b16"])
def test_full_size_blocks_reach_the_phase_pipelined_kernels(name):
    """VERDICT r3 #1a: the 256-channel residual blocks at B = 16, 64 x 64 -- the oracle / fixture comparison of
    test_hip_matches_oracle_and_fixture above is only worth its name if the kernels the STEP selects at this shape are
    the ones that ran: the phase-pipelined igemm (forward with the InstanceNorm-partial epilogue, data gradient with the
    reflect fold and the residual gradient of BlockLink) and the phase-pipelined weight gradient."""
    got, names = _timed_case(name, "bf16")
    assert "conv_igemm_p8<bf16,256x256>" in names, names
    assert "conv_wgrad_p8<bf16,256x256>" in names, names
    _check(name, got, _oracle(name), "bf16", "oracle")


# cases small enough for the CPU oracle whose layers the full-size step would hand to the phase-pipelined igemm
# (Co > 128) and the halo-tile kernel (3 x 3, Co 64 / 128, rows of a multiple of 32 pixels) if only they had >= 256
# (512) tiles: with the tile-count threshold lowered (o2m_debug_fill_blocks, a test hook of the C ABI) they ARE
# handed to them -- partial tiles, tiny grids and per-sample filters included
FORCED_OP_CASES = ["gen64_deep"]


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("name", FORCED_OP_CASES)
def test_small_cases_routed_to_the_big_tile_kernels_match_oracle_and_fixture(name, precision, golden_dir):
    """VERDICT r3 #1b.  bf16: the p8 / halo kernels under the oracle and the reference fixture; fp32: the same routing
    decisions taken for the fp32-split tiles (256 x 256 symmetric kernel instead of 128 x 128)."""
    from one_to_many_gan_amd import _hip as H

    prev = H.debug_fill_blocks(1)
    try:
        got, names = _timed_case(name, precision)
    finally:
        H.debug_fill_blocks(prev)
    if precision == "bf16":
        assert {"conv_igemm_p8<bf16,256x256>", "conv3x3_halo<bf16,8x32x64>", "conv3x3_halo<bf16,8x32x128>"} <= names, names
    _check(name, got, _oracle(name), precision, "oracle")
    gold = np.load(os.path.join(golden_dir, f"{name}.npz"))
    _check(name, got, {k: gold[k] for k in gold.files}, precision, "reference fixture")


SHARED_MASK_CASES = [n for n in OP_CASES if n.startswith(NET_CASES) or n.endswith("_c256_b16")]
# Measured on MI355X (tools/parity_report.py --shared-masks, profiles/r02_shared_masks_report.txt): with the
# masks shared, the fp32-mode gradients sit this far from the fp64 oracle, per case (worst tensor).
SHARED_MASK_TOL = 1e-3


@pytest.mark.parametrize("name", SHARED_MASK_CASES)
def test_net_gradients_with_shared_activation_masks(name):
    """Net-level gradients held to the op-level bound once mask-flip noise is taken out
    (tests/cases.py::run_case_shared_masks): fp32 parity mode vs the FLOAT64 oracle replaying the HIP
    path's own ReLU / LeakyReLU masks.  A kernel defect confined to the composed paths -- a filter's
    gradient accumulated over several uses, the side-stream style backward, the 2B discriminator
    pass, tile selection at Co > 256 -- shows up here; a flipped mask does not."""
    from tests.cases import run_case_shared_masks

    got, want, flipped, total = run_case_shared_masks(name, product_ns("fp32"), oracle_ns())
    assert total > 0 and flipped <= 1e-3 * total, (flipped, total)  # the forward error is ~1e-5
    bad = []
    for k, w in want.items():
        if k.endswith("/sum") or k.endswith("/sqsum") or float(w.abs().max()) == 0:
            continue
        err = _rel(got[k], w)
        if err > SHARED_MASK_TOL:
            bad.append((k, err))
    assert not bad, (name, flipped, total, bad)


# Post-step probe bounds of the cases whose probes mostly measure Adam's sign(g) on noise-level gradients.
# steps128_stock = steps128 (config #4's topology, 36 M-parameter generator) at the stock lr 2e-3: Adam's first update is
# +-lr per weight, so the fraction of weights whose tiny gradient rounds the other way moves the probe images by an
# amount proportional to lr.  Measured round 2 (gpurun_out/r02/gputest1.log): probe/img 6.6e-2 in fp32 mode and 0.49 in
# bf16; at lr 2e-5 (steps128, default bounds 5e-2 / 2e-1) the same code measures 1e-3 / 2e-2.  The logged losses
# of BOTH steps stay on the default bounds in both cases.
PROBE_TOL = {"steps128_stock": {"fp32": 0.2, "bf16": 0.7}}


def _check_steps(name, got, gold, precision):
    """Two consecutive D+G steps (Adam included) against the reference's fixture: logged losses and post-step probes."""
    # step 0 depends on the forward only; step 1 and the probes also on one Adam update of
    # every parameter by ~lr*sign(g) -- sign flips of tiny gradients perturb them slightly
    # (bf16: gradient noise flips the sign of ~1/6 of the +-lr Adam moves, see the yardstick)
    # Adam's first update is lr*sign(g): every weight whose |g| lies below the gradient noise
    # floor (ReLU-mask flips already put ~1e-2 of relative noise on encoder gradients in fp32
    # mode) moves by +-2 lr, whatever the precision
    loose = {"fp32": 5e-2, "bf16": 2e-1}[precision]
    tight = {"fp32": 2e-3, "bf16": 5e-2}[precision]
    probe_tol = PROBE_TOL.get(name, {}).get(precision, loose)
    bad = []
    for k in gold.files:
        w = torch.from_numpy(gold[k])
        if k.endswith("/sum"):
            continue
        if k.startswith("step"):
            # compare each logged scalar on the scale of the step's largest loss term
            err = float((got[k] - w).abs().max() / w.abs().max())
            tol = tight if k.startswith("step0") else loose
            if k.endswith("/d"):
                # confidences are means of sign(): a single flipped patch moves them by 2/N
                err = float((got[k][:1] - w[:1]).abs().max() / w[:1].abs().max())
        else:
            err, tol = _rel(got[k], w), probe_tol
        if err > tol:
            bad.append((k, err, tol))
    assert not bad, bad


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", STEP_CASES)
def test_training_steps_match_reference(name, precision, golden_dir):
    """Two consecutive D+G steps (Adam included): logged losses and post-step probes."""
    got = run_case(name, product_ns(precision), "cuda")
    _check_steps(name, got, np.load(os.path.join(golden_dir, f"{name}.npz")), precision)


def test_training_steps_routed_to_the_big_tile_kernels_match_reference(golden_dir):
    """VERDICT r3 #1b on the whole step: steps128 (config #4's topology: 512-channel latent at 16 x 16, B = 2) with
    the tile-count threshold lowered, so that both step functions run on the phase-pipelined igemm (plain, per-sample
    filters, InstanceNorm partials, reflect fold) and the halo-tile kernel (style-dot epilogue included) -- the kernels
    of the 256 x 256, B = 16 step -- and still reproduce the reference's two steps."""
    from one_to_many_gan_amd import _hip as H

    prev = H.debug_fill_blocks(1)
    try:
        got, names = _timed_case("steps128", "bf16")
    finally:
        H.debug_fill_blocks(prev)
    assert {"conv_igemm_p8<bf16,256x256>", "conv3x3_halo<bf16,8x32x64>", "conv3x3_halo<bf16,8x32x128>"} <= names, names
    _check_steps("steps128", got, np.load(os.path.join(golden_dir, "steps128.npz")), "bf16")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_deterministic_mode_makes_training_steps_bitwise_reproducible(precision):
    """The reference's ``deterministic_cuda_kernels`` switch (train.py:41-45): with
    ``set_deterministic(True)`` two runs of the same two D+G steps (Adam included) agree bit for bit --
    losses and the post-step probes of all four networks."""
    import one_to_many_gan_amd as pk

    pk.set_deterministic(True)
    try:
        a = run_case("steps64", product_ns(precision), "cuda")
        b = run_case("steps64", product_ns(precision), "cuda")
    finally:
        pk.set_deterministic(False)
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, diff


@pytest.mark.gpu
def test_discriminator_step_on_its_own_stream_gives_the_same_steps():
    """ops.d_step_stream (round 4): discriminator_step queued on a stream of its own, the generator step that follows
    ordered behind it only where it uses the discriminator.  Scheduling only: two D+G steps (Adam included) with and
    without it agree to the run-to-run noise of the non-deterministic mode (the streams exist only there; its
    atomically accumulated sums make two runs of the SAME configuration differ by up to 6e-3 in the noisiest probe
    after two Adam steps -- tools/overlap_noise.py: off / off 5e-4, on / on 6e-3, one stream 5.6e-3, off / on 1e-3 and
    5e-3), and the overlapped run itself matches the reference fixtures (test_training_steps_match_reference runs with
    the default, i.e. with the overlap)."""
    from one_to_many_gan_amd import ops

    assert ops._D_OVERLAP, "the default configuration runs the discriminator step on its own stream"
    with_overlap = run_case("steps64", product_ns("fp32"), "cuda")
    assert ops._DSTREAM, "the discriminator-step stream was used"
    ops._D_OVERLAP = False
    try:
        without = run_case("steps64", product_ns("fp32"), "cuda")
    finally:
        ops._D_OVERLAP = True
    for k in with_overlap:
        a, b = with_overlap[k].double(), without[k].double()
        err = float((a - b).norm() / max(float(b.norm()), 1e-30))
        assert err < 3e-2, (k, err)


@pytest.mark.gpu
def test_config4_step_at_full_size_is_finite_and_reproducible():
    """BASELINE config #4 at its real size (512 x 512 RGB, batch 8, bf16: 3 down-samplings, 512-channel latent, the
    3B = 24 decode group at 512 x 512): one D+G step through the product's step functions.  No CPU oracle finishes
    this size, so the checks are the size-independent ones: every logged scalar finite, the probes finite and not
    degenerate, and -- deterministic mode -- two runs bit for bit equal (every kernel's tile / slice / chunk
    selection at this size included)."""
    import one_to_many_gan_amd as pk
    from tests.cases import case_steps

    pk.set_deterministic(True)
    try:
        runs = [case_steps(product_ns("bf16"), torch.device("cuda"), tag="steps512", nc=3, size=(512, 512), batch=8,
                           n_steps=1) for _ in range(2)]
    finally:
        pk.set_deterministic(False)
    a, b = runs
    for k, v in a.items():
        assert torch.isfinite(v.float()).all(), k
    assert float(a["step0/g"][0]) > 0 and float(a["step0/d"][0]) > 0
    diff = [k for k in a if not torch.equal(a[k], b[k])]
    assert not diff, diff


@pytest.mark.gpu
def test_async_logged_scalars_equal_the_blocking_read():
    """core.training.set_async_scalars(True) (train.py, bench.py): the ten logged scalars of a D+G step come back as
    LoggedScalar objects whose device->host copy is read on first use.  Deterministic mode, same seeds: their values
    are bit for bit those of the blocking read, and the Logger line built from them is the same line."""
    import one_to_many_gan_amd as pk
    from one_to_many_gan_amd.core import training as pt
    from one_to_many_gan_amd.core.evaluation import Logger

    pk.set_deterministic(True)
    try:
        sync = run_case("steps64", product_ns("bf16"), "cuda")
        pt.set_async_scalars(True)
        lazy = run_case("steps64", product_ns("bf16"), "cuda")
    finally:
        pt.set_async_scalars(False)
        pk.set_deterministic(False)
    diff = [k for k in sync if not torch.equal(sync[k], lazy[k])]
    assert not diff, diff

    pending = pt._PendingScalars(torch.tensor([1.5, -2.0, 0.25], device="cuda"))
    a, b, c = (pt.LoggedScalar(pending, i) for i in range(3))
    assert pending.values is None  # nothing read yet
    assert float(a) == 1.5 and a + 1 == 2.5 and 1 + a == 2.5 and a * b == -3.0 and abs(b) == 2.0 and -c == -0.25
    assert f"{c:.3f}" == "0.250" and b < a and sum([a, b, c]) == -0.25 and np.isfinite(float(c))
    lg_lazy, lg_plain = Logger(4), Logger(4)
    for lg, vals in ((lg_lazy, (a, b, c)), (lg_plain, (1.5, -2.0, 0.25))):
        for attr in vars(lg):
            if attr.startswith("log_"):
                getattr(lg, attr).extend(vals)
    assert lg_lazy.print(4) == lg_plain.print(4)
