"""GPU tests of the linear-channel VAE drop-ins (reference LinearVAE / SecondVolterraVAE, lib/real_equalization.py:564-736)
against golden vectors produced by the reference (tests/golden/make_golden_linear_channel.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-5  # fp32 relative tolerance of the hot path (20-step trajectories add the reference's own fp32-vs-fp64 distance)


def rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make(g):
    from volterra_vae_amd.real_equalization import LinearVAE, SecondVolterraVAE
    sched = str(g["lr_schedule"]) if "lr_schedule" in g.files else "step"
    kw = dict(learning_rate=float(g["lr"]) if "lr" in g.files else 5e-3, constellation=g["constellation"],
              samples_per_symbol=int(g["S"]), batch_size=int(g["B"]), noise_variance=float(g["noise_variance_in"]),
              torch_device=torch.device("cuda"), lr_schedule=None if sched == "none" else sched)
    if str(g["kind"]) == "linvae":
        return LinearVAE(channel_n_taps=int(g["L"]), equaliser_n_taps=int(g["l1"]), **kw)
    return SecondVolterraVAE(channel_memory=int(g["L"]), equaliser_n_lags1=int(g["l1"]), equaliser_n_lags2=int(g["l2"]), **kw)


@pytest.mark.parametrize("case", ["linvae", "svvae", "svvae_L8S1"])
def test_single_step_loss_and_gradients(case):
    g32 = np.load(os.path.join(GOLDEN, f"vae_real_step_{case}_f32.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"vae_real_step_{case}_f64.npz"))
    m = make(g32)
    assert repr(m) == ("LinVAE" if str(g32["kind"]) == "linvae" else "SecondVolterraVAE")
    assert [p["name"] for p in m.get_parameters()] == ["channel_filter", "equaliser"]
    with torch.no_grad():
        m.channel_filter.copy_(torch.from_numpy(g32["h"]))
        if str(g32["kind"]) == "linvae":
            m.equaliser.weight.copy_(torch.from_numpy(g32["eq_w"]))
        else:
            m.equaliser.kernel1.copy_(torch.from_numpy(g32["k1"]))
            m.equaliser.kernel2.copy_(torch.from_numpy(g32["k2"]))
    y = torch.from_numpy(g32["rx"]).float().cuda()
    xhat = m.forward(y)
    assert rel(xhat, g64["xhat"]) < TOL
    q = m._soft_demapping(xhat)
    assert rel(q, g64["q"]) < TOL
    loss = m._calculate_loss(xhat, y)
    assert abs(float(loss.detach()) - float(g64["loss"])) < TOL * abs(float(g64["loss"]))
    assert abs(float(m.noise_variance) - float(g64["noise_variance_out"])) < TOL * float(g64["noise_variance_out"])
    loss.backward()

    def chk(ours, key):
        # 1e-5 against the reference's fp64 run with no allowance; against its fp32 run the reference's own fp32 - fp64 distance
        # is added
        own = rel(g32[key], g64[key])
        e64, e32 = rel(ours, g64[key]), rel(ours, g32[key])
        print(f"{case} {key}: vs fp64 ref {e64:.2e}, vs fp32 ref {e32:.2e} (ref32-ref64 {own:.2e})")
        assert e64 < TOL, (key, e64)
        assert e32 < TOL + own, (key, e32, own)
    chk(m.channel_filter.grad, "g_h")
    if str(g32["kind"]) == "linvae":
        chk(m.equaliser.weight.grad, "g_eq_w")
    else:
        chk(m.equaliser.kernel1.grad, "g_k1")
        chk(m.equaliser.kernel2.grad, "g_k2")
    assert m.channel_h2.grad is None  # the quadratic channel kernel is not a parameter of these models


@pytest.mark.parametrize("case", ["linvae", "svvae"])
@pytest.mark.parametrize("persistent", [None, False])
def test_fit_trajectory_matches_reference(case, persistent):
    g32 = np.load(os.path.join(GOLDEN, f"vae_real_fit_{case}_f32.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"vae_real_fit_{case}_f64.npz"))
    m = make(g32)
    m.persistent_fit = persistent
    m.initialize_optimizer()
    y_eq, loss_curve = m.fit(g32["rx"])
    assert float(m.channel_h2.abs().max()) == 0.0 and float((m.noise_scaling - 1).abs().max()) == 0.0

    def chk(ours, key, tol):
        own = rel(g32[key], g64[key])
        assert rel(ours, g64[key]) < tol + own, (key, rel(ours, g64[key]), own)
    chk(loss_curve, "loss_curve", TOL)
    chk(y_eq, "y_eq", TOL)
    chk(m.channel_filter, "h", TOL)
    if case == "linvae":
        chk(m.equaliser.weight, "eq_w", TOL)
    else:
        chk(m.equaliser.kernel1, "k1", TOL)
        chk(m.equaliser.kernel2, "k2", 1e-4)
    assert abs(float(m.noise_variance) - float(g64["noise_variance_out"])) < TOL * float(g64["noise_variance_out"])
    q = m.estimate_symbol_probs(g32["rx_val"])
    assert rel(q, g32["q_val"]) < 1e-4
    assert (q.argmax(axis=0) == g32["q_val"].argmax(axis=0)).mean() > 0.999
