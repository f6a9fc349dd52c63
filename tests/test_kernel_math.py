"""The hand-derived fused forward/backward (numpy emulation of the kernel arithmetic) against
the golden vectors (real path) and against torch autograd on the oracle (complex path)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import kernel_math as km
from oracle import v2vae_oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
STEP64 = sorted(os.path.basename(p)[len("v2vae_real_step_"):-4]
                for p in glob.glob(os.path.join(GOLDEN, "v2vae_real_step_*f64.npz")))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("name", STEP64)
def test_real_fused_math_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name}.npz"))
    S = int(g["S"])
    levels = g["constellation"][None, :]
    prior = np.ones_like(levels) / levels.shape[1]
    if int(g["linear_eq"]):
        k1, k2 = g["eq_w"].reshape(-1), None
        pad1 = (len(k1) - 1) // 2
    else:
        k1, k2, pad1 = g["k1"], g["k2"], None
    out = km.full_step_grads(0, g["rx"][None, :], (g["h1"], g["h2"], k1, k2), levels, prior,
                             g["noise_scaling"][None, :], float(g["noise_variance_in"]), S, pad1)
    assert rel(out["xhat"][0], g["xhat"]) < 1e-12
    assert rel(out["q"][0], g["q"]) < 1e-10
    assert abs(out["loss"] - float(g["loss"])) < 1e-10 * abs(float(g["loss"]))
    assert abs(out["E"] - float(g["E"])) < 1e-10 * abs(float(g["E"]))
    assert rel(out["g_h1"], g["g_h1"]) < 1e-9
    assert rel(out["g_h2"], g["g_h2"]) < 1e-9
    if not int(g["no_scaling"]):
        assert rel(out["g_scaling"][0], g["g_noise_scaling"]) < 1e-9
    if int(g["linear_eq"]):
        assert rel(out["g_k1"], g["g_eq_w"].reshape(-1)) < 1e-9
    else:
        assert rel(out["g_k1"], g["g_k1"]) < 1e-9
        assert rel(out["g_k2"], g["g_k2"]) < 1e-9


def _complex_case(seed, L, S, l1, l2, B, M):
    rng = np.random.default_rng(seed)
    N = B * S
    y = rng.standard_normal((2, N))
    lv = np.arange(-(M - 1), M, 2).astype(float)
    levels = np.stack((lv, lv))
    prior = np.ones_like(levels) / M
    scaling = 1.0 + 0.2 * rng.random((2, M))
    h1 = 0.2 * rng.standard_normal((2, L)); h1[0, L // 2] += 1.0
    h2 = 0.05 * rng.standard_normal((2, L, L))
    k1 = 0.1 * rng.standard_normal((2, l1)); k1[0, l1 // 2] += 1.0
    k2 = 0.02 * rng.standard_normal((2, l2, l2)) if l2 else None
    return y, levels, prior, scaling, h1, h2, k1, k2


@pytest.mark.parametrize("L,S,l1,l2,B,M", [(7, 2, 5, 3, 40, 4), (6, 1, 3, 0, 30, 2), (9, 3, 7, 5, 33, 4)])
def test_complex_fused_math_matches_autograd(L, S, l1, l2, B, M):
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_case(5, L, S, l1, l2, B, M)
    nv = 0.3
    out = km.full_step_grads(1, y, (h1, h2, k1, k2), levels, prior, scaling, nv, S)
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tsc = t(h1), t(h2), t(k1), t(scaling)
    tk2 = t(k2) if k2 is not None else None
    ty = torch.tensor(y)
    xhat = orc.complex_equaliser(ty, tk1[0], tk1[1], None if tk2 is None else tk2[0], None if tk2 is None else tk2[1], S)
    loss, E, kl = orc.complex_v2vae_loss(xhat, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels),
                                         torch.tensor(prior), torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    assert rel(out["xhat"], xhat.detach().numpy()) < 1e-12
    assert abs(out["loss"] - float(loss.detach())) < 1e-10 * abs(float(loss.detach()))
    assert rel(out["g_h1"], th1.grad) < 1e-9
    assert rel(out["g_h2"], th2.grad) < 1e-9
    assert rel(out["g_k1"], tk1.grad) < 1e-9
    assert rel(out["g_scaling"], tsc.grad) < 1e-9
    if tk2 is not None:
        assert rel(out["g_k2"], tk2.grad) < 1e-9


def test_owned_range_partition_sums_to_whole():
    """Halo-sharding contract: E and parameter partials over disjoint owned ranges add up, and the
    per-symbol gradients of each owned range are complete."""
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_case(9, 7, 2, 5, 3, 48, 4)
    S, L, B = 2, 7, 48
    pre, end = km.delays(L)
    K1s, K2s = km.stacked_equaliser_kernels(1, k1, k2)
    xhat = km.equaliser_forward(y, K1s, K2s, S, 5 // 2, B)
    q, mom, kl = km.demap_moments(xhat, levels, prior, scaling, 0.3, pre, B + end)
    tabs = km.polyphase_tables(*km.stacked_channel(1, h1, h2), L, S)
    E, gmom, parts = km.elbo_fused(y, mom, tabs, L, S)
    Es, gm = 0.0, np.zeros_like(gmom)
    acc = {k: np.zeros_like(v) for k, v in parts.items()}
    for lo, hi in ((0, 17), (17, 30), (30, 48)):
        e, g, p = km.elbo_fused(y, mom, tabs, L, S, own=(lo, hi))
        Es += e
        gm[:, :, lo:hi] = g[:, :, lo:hi]
        for k in acc:
            acc[k] += p[k]
    assert abs(Es - E) < 1e-10 * abs(E)
    assert rel(gm, gmom) < 1e-12
    for k in acc:
        assert rel(acc[k], parts[k]) < 1e-10
