"""Seeded synthetic nonlinear-ISI data for tests and benchmarks (host side, numpy).

The reference's own simulator (lib/data_generation.py:318-390, `NonLinearISI`) needs
scikit-commpy, which is not available offline, so throughput and parity runs use this
self-contained generator instead: uniform PAM / square-QAM symbols, zero-stuffed to
`sps` samples per symbol, a short low-pass pulse, a Wiener-Hammerstein system
(FIR -> 2nd-order polynomial -> FIR, tap values from main_time_varying_wh.py:36-42),
normalisation to zero mean / unit standard deviation (data_generation.py:241) and AWGN.
"""
from __future__ import annotations

import numpy as np

FIR1 = (1.0, 0.3, 0.1)
FIR1_ALT = (1.0, 0.5, 0.1525)
FIR2 = (1.0, -0.2, 0.02)
POLY = (0.9, 0.1)


def pam_constellation(order: int) -> np.ndarray:
    """PAM-M amplitude levels [-(M-1), ..., M-1] in steps of 2 (what main_v2vae.py:31-34 prints)."""
    return np.arange(-(order - 1), order, 2).astype(np.float64)


def qam_constellation(order: int) -> np.ndarray:
    """Square M-QAM: cartesian product of sqrt(M)-PAM levels, complex128, row-major (I outer, Q inner)."""
    m = int(round(np.sqrt(order)))
    if m * m != order:
        raise ValueError("only square QAM is supported")
    lv = pam_constellation(m)
    return (lv[:, None] + 1j * lv[None, :]).reshape(-1)


def _lowpass_pulse(sps: int) -> np.ndarray:
    if sps == 1:
        return np.ones(1)
    span = 4
    t = np.arange(-span * sps, span * sps + 1) / sps
    pulse = np.sinc(t) * np.hamming(len(t))
    return pulse / np.max(pulse)


def nonlinear_isi(n_symbols: int, constellation: np.ndarray, sps: int, snr_db: float,
                  rng: np.random.Generator, fir1=FIR1, fir2=FIR2, poly=POLY):
    """Return (rx[n_symbols*sps], syms[n_symbols]); complex output iff the constellation is complex."""
    constellation = np.asarray(constellation)
    idx = rng.integers(0, len(constellation), size=n_symbols)
    syms = constellation[idx]
    x = np.zeros(n_symbols * sps, dtype=syms.dtype)
    x[::sps] = syms
    x = np.convolve(x, _lowpass_pulse(sps), mode="same")
    z = np.convolve(x, np.asarray(fir1), mode="same")
    w = poly[0] * z + poly[1] * z * z
    out = np.convolve(w, np.asarray(fir2), mode="same")
    out = out - np.mean(out)
    out = out / np.std(out)
    noise_std = np.sqrt(10.0 ** (-snr_db / 10.0))
    if np.iscomplexobj(out):
        noise = (rng.standard_normal(out.shape) + 1j * rng.standard_normal(out.shape)) * (noise_std / np.sqrt(2.0))
    else:
        noise = rng.standard_normal(out.shape) * noise_std
    return out + noise, syms


_NORM_CACHE = {}


def _wh_chain(x, sps, fir1, fir2, poly):
    x = np.convolve(x, _lowpass_pulse(sps), mode="same")
    z = np.convolve(x, np.asarray(fir1), mode="same")
    w = poly[0] * z + poly[1] * z * z
    return np.convolve(w, np.asarray(fir2), mode="same")


def nonlinear_isi_range(seed: int, constellation: np.ndarray, sps: int, snr_db: float, sym_lo: int, sym_hi: int,
                        b_glob: int, fir1=FIR1, fir2=FIR2, poly=POLY, block: int = 1 << 16, margin: int = 32):
    """Samples [sym_lo*sps, sym_hi*sps) of one global b_glob-symbol sequence, reproducible from
    (seed, symbol index) alone, so that ranks holding overlapping ranges see identical samples.
    Symbols and noise are drawn per `block` symbols from generators seeded by (seed, kind, block id);
    the normalisation constants come from block 0 so that they do not depend on the range."""
    constellation = np.asarray(constellation)
    cplx = np.iscomplexobj(constellation)

    def block_syms(j):
        rng = np.random.default_rng([seed, 0, j])
        return constellation[rng.integers(0, len(constellation), size=block)]

    def block_noise(j):
        rng = np.random.default_rng([seed, 1, j])
        if cplx:
            return (rng.standard_normal(block * sps) + 1j * rng.standard_normal(block * sps)) / np.sqrt(2.0)
        return rng.standard_normal(block * sps)

    key = (seed, sps, tuple(fir1), tuple(fir2), tuple(poly), constellation.tobytes(), block)
    if key not in _NORM_CACHE:
        x0 = np.zeros(block * sps, dtype=constellation.dtype)
        x0[::sps] = block_syms(0)
        o0 = _wh_chain(x0, sps, fir1, fir2, poly)
        _NORM_CACHE[key] = (np.mean(o0), np.std(o0))
    mean, std = _NORM_CACHE[key]

    lo = max(0, sym_lo - margin)
    hi = min(b_glob, sym_hi + margin)
    j0, j1 = lo // block, (hi - 1) // block
    syms = np.concatenate([block_syms(j) for j in range(j0, j1 + 1)])[lo - j0 * block: hi - j0 * block]
    x = np.zeros((hi - lo) * sps, dtype=constellation.dtype)
    x[::sps] = syms
    out = (_wh_chain(x, sps, fir1, fir2, poly) - mean) / std
    out = out[(sym_lo - lo) * sps: (sym_hi - lo) * sps]
    n0, n1 = sym_lo // block, (sym_hi - 1) // block
    noise = np.concatenate([block_noise(j) for j in range(n0, n1 + 1)])[(sym_lo - n0 * block) * sps: (sym_hi - n0 * block) * sps]
    return out + noise * np.sqrt(10.0 ** (-snr_db / 10.0)), syms[sym_lo - lo: sym_hi - lo]
