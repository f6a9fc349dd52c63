"""Host-side metrics with the semantics of lib/utility.py (pure numpy/scipy, no GPU).

Only what the post-fit symbol-error-rate check needs: arg-max decoding of posteriors (:60-69),
delay search by full cross-correlation of the real parts (:40-45) and the SER with `discard`
symbols trimmed at both ends (:72-87).
"""
from __future__ import annotations

import numpy as np
from scipy.signal import correlate


def find_delay(x, y):
    """Lag that maximises the cross-correlation of the real parts (lib/utility.py:40-45)."""
    return int(np.argmax(correlate(np.real(x), np.real(y)))) - x.shape[0] + 1


def decode_from_probs(q, constellation):
    """q [M, N] posteriors -> most likely symbol per time step (lib/utility.py:60-69)."""
    assert q.shape[0] == len(constellation)
    return np.asarray(constellation)[np.argmax(q, axis=0)]


def decode_from_indices(idx, constellation):
    """Hard decisions given as level indices (e.g. from v2vae_decide) -> symbols."""
    return np.asarray(constellation)[np.asarray(idx)]


def ser_from_decisions(ahat, a, discard=10):
    opt_delay = find_delay(ahat, a)
    errors = ahat[discard:-discard] != np.roll(a, opt_delay)[discard:-discard]
    return float(np.mean(errors)), opt_delay


def calc_ser_from_probs(q, a, constellation, discard=10):
    """lib/utility.py:72-77 (including its progress print)."""
    ahat = decode_from_probs(q, constellation)
    opt_delay = find_delay(ahat, a)
    errors = ahat[discard:-discard] != np.roll(a, opt_delay)[discard:-discard]
    print(f"Total number of erros {np.sum(errors)} (optimal delay: {opt_delay})")
    return np.mean(errors), opt_delay


def decision_logic(xhat, syms, symbol_centers=None):
    """Nearest-centre hard decision (lib/utility.py:163-171)."""
    symbol_centers = syms if symbol_centers is None else symbol_centers
    idx = np.argmin(np.abs(np.asarray(xhat)[:, None] - np.asarray(symbol_centers)[None, :]), axis=1)
    return np.asarray(syms)[idx]


def calc_ser_pam(y_eq, a, discard=10):
    """lib/utility.py:80-87."""
    assert len(y_eq) == len(a)
    opt_delay = find_delay(y_eq, a)
    const = np.unique(a)
    ahat = decision_logic(y_eq, const, const)
    errors = ahat[discard:-discard] != np.roll(a, opt_delay)[discard:-discard]
    print(f"Total number of erros {np.sum(errors)} (optimal delay: {opt_delay})")
    return np.mean(errors), opt_delay
