"""Synthetic nonlinear-ISI sequences produced on the GPU (v2vae_synth_channel): the device-side twin of `synth.nonlinear_isi`.

Same chain and filter values as the host generator (pulse, FIR1 -> polynomial -> FIR2, normalisation, AWGN), but symbols and
noise come from a counter-based generator keyed by (seed, global index), so a rank can produce exactly its own sample range of a
2^26-symbol sequence in HBM without any host work or PCIe traffic, and overlapping ranges agree bit for bit."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import synth
from ._lib import SynthArgs, check, lib

_NORM = {}


def _args(seed, levels, ns, sps, b_glob, sym_lo, sym_cnt, fir1, fir2, poly, mean, inv_std, noise_std):
    a = SynthArgs()
    pulse = synth._lowpass_pulse(sps)
    a.ns, a.sps, a.C, a.n_pulse, a.n_fir1, a.n_fir2 = ns, sps, len(levels), len(pulse), len(fir1), len(fir2)
    for name, dst, src in (("levels", a.levels, levels), ("pulse", a.pulse, pulse), ("fir1", a.fir1, fir1), ("fir2", a.fir2, fir2)):
        if len(src) > len(dst):
            raise ValueError(f"{name}: at most {len(dst)} values (got {len(src)})")
        for i, v in enumerate(src):
            dst[i] = float(v)
    a.poly1, a.poly2 = float(poly[0]), float(poly[1])
    a.mean[0], a.mean[1] = float(mean[0]), float(mean[1])
    a.inv_std, a.noise_std = float(inv_std), float(noise_std)
    a.seed, a.b_glob, a.sym_lo, a.sym_cnt = seed, b_glob, sym_lo, sym_cnt
    return a


def _run(a, device, want_idx):
    y = torch.empty((a.ns, a.sym_cnt * a.sps), dtype=torch.float32, device=device)
    idx = torch.empty((a.ns, a.sym_cnt), dtype=torch.int32, device=device) if want_idx else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.v2vae_synth_channel(C.byref(a), C.c_void_p(y.data_ptr()), None if idx is None else C.c_void_p(idx.data_ptr()),
                                  stream), "v2vae_synth_channel")
    return y, idx


def nonlinear_isi_device(seed: int, levels, ns: int, sps: int, snr_db: float, b_glob: int, sym_lo: int = 0, sym_cnt=None,
                         device="cuda", fir1=synth.FIR1, fir2=synth.FIR2, poly=synth.POLY, norm_symbols: int = 1 << 16,
                         return_indices: bool = False):
    """Samples [sym_lo * sps, (sym_lo + sym_cnt) * sps) of one global b_glob-symbol sequence as a float32 tensor [ns, sym_cnt * sps]
    on `device` (ns = 1: PAM with amplitude `levels`; ns = 2: square QAM, rows I and Q), zero mean / unit power before noise, AWGN
    at `snr_db`.  The normalisation constants are measured once per configuration on the noiseless first `norm_symbols` symbols,
    so they do not depend on the range.  With return_indices also the level indices [ns, sym_cnt] (int32) of the symbols sent."""
    sym_cnt = b_glob - sym_lo if sym_cnt is None else sym_cnt
    levels = [float(v) for v in levels]
    key = (seed, tuple(levels), ns, sps, b_glob, tuple(fir1), tuple(fir2), tuple(poly), norm_symbols, str(device))
    if key not in _NORM:
        n0 = min(norm_symbols, b_glob)
        raw, _ = _run(_args(seed, levels, ns, sps, b_glob, 0, n0, fir1, fir2, poly, (0.0, 0.0), 1.0, 0.0), device, False)
        raw = raw.double()
        mean = raw.mean(dim=1)
        std = torch.sqrt(((raw - mean[:, None]) ** 2).sum() / raw.shape[1])  # complex: total power of both streams
        _NORM[key] = ([float(v) for v in mean] + [0.0], float(std))
    mean, std = _NORM[key]
    noise_std = np.sqrt(10.0 ** (-snr_db / 10.0)) / (np.sqrt(2.0) if ns == 2 else 1.0)
    y, idx = _run(_args(seed, levels, ns, sps, b_glob, sym_lo, sym_cnt, fir1, fir2, poly, mean, 1.0 / std, noise_std), device,
                  return_indices)
    return (y, idx) if return_indices else y
