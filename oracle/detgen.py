"""Framework-independent deterministic tensor generator (test infrastructure).

Golden fixtures cannot carry 14 M fp32 weights, so weights and inputs are GENERATED on both sides
from a counter-based integer hash (splitmix64 over the element index, keyed by crc32 of the tensor
name): the fixture generator (tests/golden/make_golden.py, run in the build container against the
real reference) and the tests (run anywhere) call the same function and get bit-identical float32
values.  Pure numpy integer arithmetic — no RNG state, no torch version dependence.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(idx: np.ndarray, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0, salt: int = 0) -> np.ndarray:
    """float32 array of `shape`, uniform in [lo, hi), fully determined by (name, salt)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = (zlib.crc32(name.encode()) + 0x1000193 * salt) & 0xFFFFFFFF
    bits = _splitmix(np.arange(n, dtype=np.uint64), seed)
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # 24-bit mantissa -> exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(name: str, shape, salt: int = 0) -> np.ndarray:
    """Zero-mean, unit-variance-ish values (sum of 3 uniforms), float32."""
    a = uniform(name, shape, -1, 1, salt * 3 + 0).astype(np.float64)
    b = uniform(name, shape, -1, 1, salt * 3 + 1).astype(np.float64)
    c = uniform(name, shape, -1, 1, salt * 3 + 2).astype(np.float64)
    return (a + b + c).astype(np.float32)


def fill_state_dict(keys_shapes, salt: int = 0):
    """Deterministic values for a state_dict given as [(key, shape, dtype_str)].

    Rules (by key suffix / rank), chosen to keep activations O(1) through deep stacks:
      *.weight rank>=2 : uniform(+-sqrt(3/fan_in))          (variance-preserving)
      norm-like weights (rank 1 '.weight'): 1 + 0.1 u
      *.bias / pe_* / rank-1 others: 0.1 u
      running_mean: 0.1 u ; running_var: 1 + 0.5 |u| ; num_batches_tracked: 0
    """
    out = {}
    for key, shape, dt in keys_shapes:
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros(shape, dtype=np.int64)
        elif key.endswith("running_var"):
            out[key] = (1.0 + 0.5 * np.abs(uniform(key, shape, salt=salt))).astype(np.float32)
        elif key.endswith("running_mean"):
            out[key] = (0.1 * uniform(key, shape, salt=salt)).astype(np.float32)
        elif len(shape) >= 2 and "pe_" not in key:
            fan_in = int(np.prod(shape[1:]))
            out[key] = (np.sqrt(3.0 / fan_in) * uniform(key, shape, salt=salt)).astype(np.float32)
        elif key.endswith(".weight") and len(shape) == 1:
            out[key] = (1.0 + 0.1 * uniform(key, shape, salt=salt)).astype(np.float32)
        else:
            out[key] = (0.1 * uniform(key, shape, salt=salt)).astype(np.float32)
    return out
