"""CPU restatement of the device-side synthetic channel (v2vae_synth_channel).  TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package imports this module.  It restates, in numpy float64, the chain the HIP kernel runs in float32:
counter-based symbols and noise (splitmix64 of (seed, kind, index)), zero-stuffing, 'same' convolutions with the pulse and the
two FIR filters around a second-order polynomial, normalisation, AWGN.  The chain is the one of this package's host generator
(volterra_vae_amd/synth.py) and, in structure, of the reference's simulator (lib/data_generation.py:206-241: `np.convolve(...,
mode='same')` twice around the non-linearity, then (z - mean) / std; :349-372: zero-stuffed symbols through a pulse, AWGN).
PARITY UNPINNED against the reference: its simulator needs scikit-commpy (lib/data_generation.py:7), which is absent here, and
it draws from numpy's generator, which a device kernel cannot reproduce; the test of the device generator is this oracle.
"""
from __future__ import annotations

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def bits(seed: int, kind: int, idx: np.ndarray) -> np.ndarray:
    """64 random bits per index; kind 0 = symbols, 1 = noise."""
    base = _mix(np.array([np.uint64(seed) ^ (np.uint64(kind) << np.uint64(56))], dtype=np.uint64))[0]
    return _mix(base ^ np.asarray(idx, dtype=np.uint64))


def symbol_indices(seed: int, m: np.ndarray, C: int):
    """(I index, Q index) of symbols m: high / low 32 bits of the draw, scaled to [0, C)."""
    z = bits(seed, 0, m)
    hi = (z >> np.uint64(32)).astype(np.uint64)
    lo = (z & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    return ((hi * np.uint64(C)) >> np.uint64(32)).astype(np.int64), ((lo * np.uint64(C)) >> np.uint64(32)).astype(np.int64)


def gaussian_pair(seed: int, n: np.ndarray):
    """Box-Muller pair (cos branch, sin branch) for samples n from two 24-bit uniforms of the draw."""
    z = bits(seed, 1, n)
    u1 = ((z >> np.uint64(40)).astype(np.float64) + 0.5) / 16777216.0
    u2 = (((z >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float64) + 0.5) / 16777216.0
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)


def channel(seed, levels, ns, sps, b_glob, pulse, fir1, fir2, poly, mean=(0.0, 0.0), inv_std=1.0, noise_std=0.0):
    """The whole global sequence: (y [ns, b_glob * sps] float64, idx [ns, b_glob])."""
    levels = np.asarray(levels, dtype=np.float64)
    m = np.arange(b_glob, dtype=np.uint64)
    ia, ib = symbol_indices(seed, m, len(levels))
    sym = levels[ia] + (1j * levels[ib] if ns == 2 else 0.0)
    x = np.zeros(b_glob * sps, dtype=sym.dtype)
    x[::sps] = sym
    same = lambda v, h: np.convolve(v, np.asarray(h, dtype=np.float64), mode="same") if len(h) > 1 else v * h[0]
    x = same(x, pulse)
    z = same(x, fir1)
    w = poly[0] * z + poly[1] * z * z
    out = same(w, fir2)
    out = (out - (mean[0] + (1j * mean[1] if ns == 2 else 0.0))) * inv_std
    if noise_std > 0:
        g0, g1 = gaussian_pair(seed, np.arange(b_glob * sps, dtype=np.uint64))
        out = out + noise_std * (g0 + (1j * g1 if ns == 2 else 0.0))
    y = np.stack((out.real, out.imag)) if ns == 2 else np.asarray(out.real)[None, :]
    idx = np.stack((ia, ib)) if ns == 2 else ia[None, :]
    return y, idx
