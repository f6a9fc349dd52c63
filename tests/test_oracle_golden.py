"""The CPU oracle against golden vectors produced by the reference (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import v2vae_oracle as orc
from volterra_vae_amd.synth import nonlinear_isi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
STEP_CASES = sorted(os.path.basename(p)[len("v2vae_real_step_"):-4] for p in glob.glob(os.path.join(GOLDEN, "v2vae_real_step_*.npz")))
FIT_CASES = sorted(os.path.basename(p)[len("v2vae_real_fit_"):-4] for p in glob.glob(os.path.join(GOLDEN, "v2vae_real_fit_*.npz")))


def _dt(name):
    return torch.float64 if name.endswith("f64") else torch.float32


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _load_step(name):
    g = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name}.npz"))
    dt = _dt(name)
    t = lambda k: torch.from_numpy(np.asarray(g[k])).to(dt)
    return g, dt, t


@pytest.mark.parametrize("name", STEP_CASES)
def test_reference_form_matches_reference(name):
    g, dt, t = _load_step(name)
    tol = 1e-11 if dt == torch.float64 else 2e-5
    S = int(g["S"])
    y = t("rx")
    levels = t("constellation")
    prior = torch.ones_like(levels) / len(levels)
    h1, h2, scal = t("h1").requires_grad_(), t("h2").requires_grad_(), t("noise_scaling").requires_grad_()
    nv = torch.tensor(float(g["noise_variance_in"]), dtype=dt)
    if int(g["linear_eq"]):
        w = t("eq_w").reshape(-1).requires_grad_()
        xhat = orc.fir_equaliser(y, w, S)
        eq_params, eq_keys = [w], ["g_eq_w"]
    else:
        k1, k2 = t("k1").requires_grad_(), t("k2").requires_grad_()
        xhat = orc.volterra_equaliser(y, k1, k2, S)
        eq_params, eq_keys = [k1, k2], ["g_k1", "g_k2"]
    assert _rel(xhat.detach().numpy(), g["xhat"]) < tol
    q = orc.soft_demap(xhat, levels, nv, scal)
    assert _rel(q.detach().numpy(), g["q"]) < tol
    loss, E, kl = orc.elbo_reference_form(xhat, y, h1, h2, levels, prior, nv, scal, S)
    assert abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])) < tol
    assert abs(float(E) - float(g["E"])) / abs(float(g["E"])) < tol
    loss.backward()
    assert _rel(h1.grad, g["g_h1"]) < 10 * tol
    assert _rel(h2.grad, g["g_h2"]) < 10 * tol
    if not int(g["no_scaling"]):
        assert _rel(scal.grad, g["g_noise_scaling"]) < 10 * tol
    for p, k in zip(eq_params, eq_keys):
        assert _rel(p.grad.reshape(-1), np.asarray(g[k]).reshape(-1)) < 10 * tol


@pytest.mark.parametrize("name", [n for n in STEP_CASES if n.endswith("f64")])
def test_window_form_equals_reference_form(name):
    """The O(W L^2) per-window expression the kernels implement == the reference's expression."""
    g, dt, t = _load_step(name)
    S, L = int(g["S"]), int(g["L"])
    y = t("rx")
    levels = t("constellation")
    prior = torch.ones_like(levels) / len(levels)
    h1, h2, scal = t("h1").requires_grad_(), t("h2").requires_grad_(), t("noise_scaling").requires_grad_()
    nv = torch.tensor(float(g["noise_variance_in"]), dtype=dt)
    xhat = t("xhat").requires_grad_()
    H = h2 + h2.T
    loss, E, kl = orc.elbo_window_form(xhat[None], y[None], h1[None], H[None], torch.diag(h2)[None],
                                       levels[None], prior[None], nv, scal[None], S)
    assert abs(float(loss) - float(g["loss"])) / abs(float(g["loss"])) < 1e-11
    loss.backward()
    assert _rel(h1.grad, g["g_h1"]) < 1e-9
    assert _rel(h2.grad, g["g_h2"]) < 1e-9
    if not int(g["no_scaling"]):
        assert _rel(scal.grad, g["g_noise_scaling"]) < 1e-9


@pytest.mark.parametrize("name", [n for n in FIT_CASES if not n.startswith("long")])
def test_fit_trajectory_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{name}.npz"))
    dt = _dt(name)
    tol = 1e-9 if dt == torch.float64 else 5e-4
    sched = str(g["lr_schedule"])
    port = orc.RealV2VAEPort(int(g["L"]), int(g["l1"]), int(g["l2"]), g["constellation"], int(g["S"]), int(g["B"]),
                             float(g["lr"]), float(g["noise_variance_in"]), dtype=dt,
                             lr_schedule=None if sched == "none" else sched, linear_eq=bool(int(g["linear_eq"])))
    y_eq, losses, nvs = port.fit(g["rx"])
    assert _rel(losses, g["loss_curve"]) < tol
    assert _rel(nvs, g["noise_variance_traj"]) < tol
    assert _rel(y_eq, g["y_eq"]) < tol
    assert _rel(port.h1.detach(), g["h1"]) < tol
    assert _rel(port.h2.detach(), g["h2"]) < 50 * tol
    assert _rel(port.scaling.detach(), g["noise_scaling"]) < tol
    if int(g["linear_eq"]):
        assert _rel(port.k1.detach(), np.asarray(g["eq_w"]).reshape(-1)) < tol
    else:
        assert _rel(port.k1.detach(), g["k1"]) < tol
        assert _rel(port.k2.detach(), g["k2"]) < 50 * tol


def test_synth_generator_is_stable():
    g = np.load(os.path.join(GOLDEN, "v2vae_real_fit_long_cfg1_f32.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    rx, _ = nonlinear_isi(int(g["B"]) * int(g["n_batches"]), g["constellation"], int(g["S"]), 16.0, rng)
    chk = float(np.sum(rx * np.cos(np.arange(len(rx)))))
    assert abs(chk - float(g["rx_checksum"])) < 1e-6 * max(1.0, abs(chk))


def test_adam_amsgrad_reference_matches_torch():
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(17)
    p = torch.tensor(p0, requires_grad=True)
    opt = torch.optim.Adam([p], lr=3e-3, amsgrad=True)
    pn, m, v, vm = p0.copy(), np.zeros(17), np.zeros(17), np.zeros(17)
    for step in range(1, 6):
        gnp = rng.standard_normal(17)
        p.grad = torch.tensor(gnp)
        opt.step()
        pn, m, v, vm = orc.adam_amsgrad_reference(pn, gnp, m, v, vm, step, 3e-3)
        assert np.allclose(pn, p.detach().numpy(), rtol=1e-12, atol=1e-14)


LIN_STEP_CASES = [f"{n}_{t}" for n in ("linvae", "svvae", "svvae_L8S1") for t in ("f64", "f32")]


@pytest.mark.parametrize("name", LIN_STEP_CASES)
def test_linear_channel_elbo_matches_reference(name):
    """The reference's linear-channel VAEs (LinearVAE, SecondVolterraVAE): the direct restatement of their ELBO and the
    second-order ELBO with h2 = 0 and unit scaling (what the kernels run for them) both reproduce the reference."""
    g = np.load(os.path.join(GOLDEN, f"vae_real_step_{name}.npz"))
    dt = _dt(name)
    tol = 1e-9 if dt == torch.float64 else 2e-4
    y = torch.from_numpy(g["rx"]).to(dt)
    h = torch.from_numpy(g["h"]).to(dt).requires_grad_(True)
    lv = torch.from_numpy(g["constellation"]).to(dt)
    prior = torch.ones_like(lv) / len(lv)
    nv = torch.tensor(float(g["noise_variance_in"]), dtype=dt)
    S = int(g["S"])
    if str(g["kind"]) == "linvae":
        k1 = torch.from_numpy(g["eq_w"].reshape(-1)).to(dt).requires_grad_(True)
        xhat = orc.fir_equaliser(y, k1, S)
        eq = [(k1, g["g_eq_w"].reshape(-1))]
    else:
        k1 = torch.from_numpy(g["k1"]).to(dt).requires_grad_(True)
        k2 = torch.from_numpy(g["k2"]).to(dt).requires_grad_(True)
        xhat = orc.volterra_equaliser(y, k1, k2, S)
        eq = [(k1, g["g_k1"]), (k2, g["g_k2"])]
    assert _rel(xhat.detach(), g["xhat"]) < tol
    loss, E, _ = orc.linear_channel_elbo(xhat, y, h, lv, prior, nv, S)
    assert abs(float(loss.detach()) - float(g["loss"])) < tol * abs(float(g["loss"]))
    W = len(g["rx"]) - int(g["L"]) + 1
    assert abs(float(E.detach()) / W - float(g["noise_variance_out"])) < tol * float(g["noise_variance_out"])
    loss.backward()
    assert _rel(h.grad, g["g_h"]) < 10 * tol
    for p, ref in eq:
        assert _rel(p.grad, ref) < 10 * tol
    # the second-order ELBO collapses to it
    h2 = torch.zeros(len(g["h"]), len(g["h"]), dtype=dt)
    loss2, E2, _ = orc.elbo_reference_form(xhat.detach(), y, h.detach(), h2, lv, prior, nv, torch.ones_like(lv), S)
    assert abs(float(loss2) - float(loss.detach())) < max(tol, 1e-12) * abs(float(loss.detach()))


POLY_STEP_CASES = [f"{n}_{t}" for n in ("mpvae", "mpvae_L8S1", "mpvae_L10S3", "hvae", "hvae_L8S1") for t in ("f64", "f32")]


def _poly_channel(g, dt):
    """(leaf tensors, a, b, expected gradients) of a polynomial-channel golden: a, b are the per-tap coefficients of x and x^2."""
    if str(g["kind"]) == "mpvae":
        h1 = torch.from_numpy(g["h1"]).to(dt).requires_grad_(True)
        h2 = torch.from_numpy(g["h2"]).to(dt).requires_grad_(True)
        return [h1, h2], h1, h2, [g["g_h1"], g["g_h2"]]
    h = torch.from_numpy(g["h"]).to(dt).requires_grad_(True)
    a1 = torch.tensor(float(g["alpha1"]), dtype=dt, requires_grad=True)
    a2 = torch.tensor(float(g["alpha2"]), dtype=dt, requires_grad=True)
    return [h, a1, a2], a1 * h, a2 * h, [g["g_h"], g["g_alpha1"], g["g_alpha2"]]


@pytest.mark.parametrize("name", POLY_STEP_CASES)
def test_polynomial_channel_elbo_matches_reference(name):
    """Memory-polynomial and Hammerstein VAEs (SecondVolterraMPVAE, SecondVolterraHVAE): the direct restatement of their
    ELBO reproduces the reference, and so does the window-form second-order ELBO with the DIAGONAL effective kernel
    H = diag(b) and the exact diagonal in the mean (what the kernels run for these models)."""
    g = np.load(os.path.join(GOLDEN, f"poly_real_step_{name}.npz"))
    dt = _dt(name)
    tol = 1e-9 if dt == torch.float64 else 2e-4
    if dt == torch.float64 and str(g["kind"]) == "hvae":
        # the reference keeps alpha_1, alpha_2 as float32 scalars whatever the model dtype (lib/real_equalization.py:1057-1058),
        # so its products alpha_1*alpha_2, alpha_1**2, alpha_2**2 (:1160) are rounded to float32 even in the float64 run
        tol = 2e-7
    y = torch.from_numpy(g["rx"]).to(dt)
    lv = torch.from_numpy(g["constellation"]).to(dt)
    prior = torch.ones_like(lv) / len(lv)
    nv = torch.tensor(float(g["noise_variance_in"]), dtype=dt)
    sc = torch.from_numpy(g["noise_scaling"]).to(dt).requires_grad_(True)
    S = int(g["S"])
    k1 = torch.from_numpy(g["k1"]).to(dt).requires_grad_(True)
    k2 = torch.from_numpy(g["k2"]).to(dt).requires_grad_(True)
    xhat = orc.volterra_equaliser(y, k1, k2, S)
    assert _rel(xhat.detach(), g["xhat"]) < tol
    leaves, a, b, expected = _poly_channel(g, dt)
    loss, E, _ = orc.polynomial_channel_elbo(xhat, y, a, b, lv, prior, nv, sc, S)
    assert abs(float(loss.detach()) - float(g["loss"])) < tol * abs(float(g["loss"]))
    W = len(g["rx"]) - int(g["L"]) + 1
    assert abs(float(E.detach()) / W - float(g["noise_variance_out"])) < tol * float(g["noise_variance_out"])
    loss.backward()
    for p, ref in zip(leaves + [k1, k2, sc], expected + [g["g_k1"], g["g_k2"], g["g_noise_scaling"]]):
        assert _rel(p.grad, ref) < 10 * tol
    # window form with a diagonal second-order kernel
    if dt == torch.float64:
        grads_direct = [p.grad.clone() for p in leaves]
        for p in leaves:
            p.grad = None
        leaves2, a2, b2, _ = leaves, *(_poly_channel_ab(g, leaves)), None
        H = torch.diag(b2)[None]
        lossw, Ew, _ = orc.elbo_window_form(xhat.detach()[None], y[None], a2[None], H, b2[None], lv[None], prior[None], nv,
                                            sc.detach()[None], S)
        assert abs(float(lossw.detach()) - float(loss.detach())) < 1e-10 * abs(float(loss.detach()))
        lossw.backward()
        for p, gd in zip(leaves2, grads_direct):
            assert _rel(p.grad, gd) < 1e-9


def _poly_channel_ab(g, leaves):
    if str(g["kind"]) == "mpvae":
        return leaves[0], leaves[1]
    return leaves[1] * leaves[0], leaves[2] * leaves[0]
