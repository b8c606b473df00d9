"""Diagnostic only: reads the s_memtime segment stamps of the stamped p8 igemm build.

  mkdir -p build/variants && cd one_to_many_gan_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC \
      -munsafe-fp-atomics -DO2M_P8_STAMPS -shared conv_igemm.hip conv_wgrad.hip pointwise.hip style.hip ada.hip \
      -o ../../build/variants/p8stamp.so
  O2M_HIP_LIB=build/variants/p8stamp.so python tools/stamp_conv.py

Prints the cycles per phase segment (fill issue, waits, MFMAs, barriers) and the epilogue timeline of two waves."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Hh, Ci, Co = 64, 256, 256
dt = torch.bfloat16
x = torch.randn(B, Hh, Hh, Ci, device="cuda").to(dt)
w = (torch.randn(Co, 3, 3, Ci, device="cuda") / 48).to(dt)
y = torch.empty(B, Hh, Hh, Co, device="cuda", dtype=dt)
for _ in range(3):
    H.conv2d_fwd(x, w, y, pad=1, pad_mode=H.PAD_REFLECT, act=H.ACT_RELU)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
H.lib().o2m_debug_stamps.argtypes = [ctypes.c_void_p]
print("rc", H.lib().o2m_debug_stamps(buf))
v = list(buf)
names = ["fill issue", "vmcnt wait", "barrier->mfma", "mfma issue", "closing barrier", "lds reads (issue+return)"]
nph = 36 * 4
for g in range(2):
    a = v[g * 8: g * 8 + 6]
    print(f"wave row {g}: cycles per PHASE: " + "  ".join(f"{n} {c / nph:.0f}" for n, c in zip(names, a)) + f"  total {sum(a) / nph:.0f}")

# epilogue timeline (absolute s_memtime values of waves 0 and 4 of block 7)
ebuf = (ctypes.c_ulonglong * 24)()
H.lib().o2m_debug_estamps.argtypes = [ctypes.c_void_p]
print("rc", H.lib().o2m_debug_estamps(ebuf))
ev = list(ebuf)
labels = ["kernel start", "main loop done", "fills drained + sync", "pass0 tile written", "pass0 barrier", "pass0 read-out done",
          "pass0 closing barrier", "pass1 tile written", "pass1 barrier", "pass1 read-out done"]
for g in range(2):
    t = ev[g * 12: g * 12 + 10]
    print(f"wave row {g}: " + "  ".join(f"{n} +{t[i] - t[i - 1] if i else 0}" for i, n in enumerate(labels)))
