"""Closed-form deterministic tensors keyed by name (test infrastructure).

Golden fixtures must not store 9 M-parameter weight files, so both sides of every parity
check (reference in the build container, oracle and HIP path on the GPU box) fill their
``state_dict`` from this integer hash.  Values are exact in fp32 on any machine: a 24-bit
integer from splitmix64 scaled by a power of two, then one fp32 multiply.
"""

from __future__ import annotations

import zlib

import numpy as np
import torch

_SQRT3 = np.float32(1.7320508)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def unit_uniform(name: str, shape, salt: int = 0) -> torch.Tensor:
    """U[0,1) fp32 tensor, a pure function of (name, salt, flat index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 random bits
    u = bits.astype(np.float32) * np.float32(2.0**-24)
    return torch.from_numpy(u.reshape(tuple(shape)))


def sym_uniform(name: str, shape, scale: float = 1.0, salt: int = 0) -> torch.Tensor:
    """Zero-mean uniform with standard deviation ``scale`` (like the N(0,1) init)."""
    u = unit_uniform(name, shape, salt)
    return (u * 2.0 - 1.0) * (_SQRT3 * np.float32(scale))


def image_batch(name: str, shape, salt: int = 0) -> torch.Tensor:
    """Synthetic image batch in [-1, 1) (the range of Normalize(0.5, 0.5))."""
    return unit_uniform(name, shape, salt) * 2.0 - 1.0


def fill_state_dict(module: torch.nn.Module, tag: str, bias_scale: float = 0.1) -> None:
    """Overwrite every parameter of ``module`` with a deterministic closed-form value.

    Weights get unit-variance uniforms (the reference initialises them N(0,1)); biases
    get small non-zero values so that bias paths are exercised; ``to_style`` biases
    stay centred on 1 as in the reference (layers.py:138-140).  Buffers (the blur
    kernels) are left alone.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            key = f"{tag}/{name}"
            if name.endswith("bias"):
                v = sym_uniform(key, p.shape, bias_scale)
                if ".to_style." in name or name.startswith("to_style."):
                    v = v + 1.0
            else:
                v = sym_uniform(key, p.shape, 1.0)
            p.copy_(v.to(p.dtype))
