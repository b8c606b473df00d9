"""Cost of the augmentation pipe at the benchmark shape (B=16, 3x256x256), p = 0.6."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import one_to_many_gan_amd as o2m

o2m.set_precision("bf16")
aug = o2m.AdaptiveDiscriminatorAugmentation(**o2m.REFERENCE_ADA_SWITCHES, generator=torch.Generator().manual_seed(0)).cuda()
aug.set_p(0.6)
x = (torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1).requires_grad_(True)
for mode in ("fwd", "fwd+bwd"):
    for _ in range(3):
        y = aug(x)
        if mode != "fwd":
            y.float().sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        y = aug(x)
        if mode != "fwd":
            y.float().sum().backward()
    torch.cuda.synchronize()
    print(f"{mode}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per call (wall, includes host sampling + operator build)")
