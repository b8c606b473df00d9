"""Stress probe: the phase-pipelined weight gradient on one stream while another stream keeps the chip busy with other
kernels (fp8 quantisation + fp8 / bf16 igemm, pointwise): its result must stay bit-identical to the one computed alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
torch.manual_seed(0)
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
partner = sys.argv[2] if len(sys.argv) > 2 else "fp8"
x = torch.randn(B, 64, 64, 256, device=dev).to(torch.bfloat16)
gy = torch.randn(B, 64, 64, 256, device=dev).to(torch.bfloat16)
def wg(out):
    out.zero_()
    H.conv2d_wgrad(x, gy, out, pad=1, pad_mode=H.PAD_REFLECT, p8=True)
ref = torch.zeros(256, 3, 3, 256, device=dev)
wg(ref); torch.cuda.synchronize()
ref2 = torch.zeros_like(ref); wg(ref2); torch.cuda.synchronize()
print("alone: repeatable", bool(torch.equal(ref, ref2)), "finite", bool(torch.isfinite(ref).all()))
# partner work
xb = torch.randn(32, 64, 64, 256, device=dev).to(torch.bfloat16)
wb = (torch.randn(256, 3, 3, 256, device=dev) / 48).to(torch.bfloat16)
yb = torch.empty(32, 64, 64, 256, device=dev, dtype=torch.bfloat16)
dq = torch.empty(2, 2, device=dev)
x8 = torch.empty(xb.shape, dtype=torch.float8_e4m3fn, device=dev); w8 = torch.empty(wb.shape, dtype=torch.float8_e4m3fn, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
bad = 0; worst = 0.0
outs = [torch.zeros_like(ref) for _ in range(4)]
for it in range(60):
    with torch.cuda.stream(sB):
        for _ in range(3):
            if partner == "fp8":
                H.quantize_fp8(xb, x8, dq[0]); H.quantize_fp8(wb, w8, dq[1])
                H.conv2d_fwd(x8, w8, yb, pad=1, pad_mode=H.PAD_REFLECT, act=H.ACT_NONE, deq=dq.view(-1))
            elif partner == "bf16":
                H.conv2d_fwd(xb, wb, yb, pad=1, pad_mode=H.PAD_REFLECT, act=H.ACT_NONE)
            else:
                yb.copy_(xb); yb.mul_(1.0001)
    with torch.cuda.stream(sA):
        o = outs[it % 4]
        wg(o)
        same = torch.equal(o, ref)
    torch.cuda.synchronize()
    if not same:
        bad += 1
        worst = max(worst, float((o - ref).abs().max()) if bool(torch.isfinite(o).all()) else float("inf"))
print(f"B={B} partner={partner}: {bad} of 60 runs differ from the result computed alone (worst abs diff {worst})")
