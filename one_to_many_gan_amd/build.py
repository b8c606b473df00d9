"""Builds the two in-tree shared objects:

* ``lib/libo2m_hip.so``   -- the hand-written HIP kernels for gfx950 behind the C ABI, with hipcc;
* ``lib/libo2m_torch.so`` -- csrc/torch_ops.cpp, the ``TORCH_LIBRARY(o2m, ...)`` operator shim over
  that ABI (host code only: g++ against the torch headers, linked to libo2m_hip.so via $ORIGIN).

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build
container; the resulting .so files travel to the GPU box with the repository snapshot.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libo2m_hip.so")
TORCH_LIB = os.path.join(HERE, "lib", "libo2m_torch.so")
SHIM = "torch_ops.cpp"
SOURCES = ["conv_igemm.hip", "conv_direct.hip", "conv_wgrad.hip", "pointwise.hip", "style.hip", "ada.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-value"]


HEADER = os.path.join(os.path.dirname(HERE), "include", "o2m_hip.h")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != SHIM]
    deps.append(HEADER)
    return any(os.path.getmtime(d) > t for d in deps)


def _shim_stale() -> bool:
    if not os.path.exists(TORCH_LIB):
        return True
    t = os.path.getmtime(TORCH_LIB)
    return any(os.path.getmtime(d) > t for d in (os.path.join(CSRC, SHIM), HEADER, LIB))


def build_shim(verbose: bool = False) -> str:
    """g++ over csrc/torch_ops.cpp: no device code, so no hipcc and no hipify pass."""
    import torch
    from torch.utils import cpp_extension as ce

    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           *[f"-I{p}" for p in ce.include_paths()], "-I/opt/rocm/include", os.path.join(CSRC, SHIM),
           "-o", TORCH_LIB + ".tmp", f"-L{os.path.dirname(LIB)}", "-lo2m_hip", f"-L{tlib}", "-lc10", "-lc10_hip",
           "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on {SHIM}:\n{r.stdout.decode(errors='replace')}")
    os.replace(TORCH_LIB + ".tmp", TORCH_LIB)
    return TORCH_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    if force or _stale():
        _build_kernels(verbose)
    if force or _shim_stale():
        build_shim(verbose)
    return LIB


def _build_kernels(verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libo2m_hip.so")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, "lib", src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
