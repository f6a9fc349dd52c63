"""Generate golden vectors from the reference implementation (runs only where /root/reference exists).

The reference (lib/real_equalization.py) is imported unmodified.  Its one unavailable
dependency, `torchaudio.functional.convolve(x, h, mode='valid')` (real_equalization.py:4 and
call sites :868-912), is replaced by a local stand-in with numpy-convolve semantics; the
stand-in is asserted equal to `np.convolve(..., 'valid')` below, which is the primitive the
reference's own derivation script uses for the same expressions
(test_volterra_expectations.py:199,226,264,346).

Output: tests/golden/v2vae_real_*.npz (inputs + expected outputs only; no reference code).
Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")


def _convolve(x, h, mode="valid"):
    assert mode == "valid"
    return F.conv1d(x[None, None, :], torch.flip(h, (0,))[None, None, :]).reshape(-1)


_ta = types.ModuleType("torchaudio")
_taf = types.ModuleType("torchaudio.functional")
_taf.convolve = _convolve
_ta.functional = _taf
sys.modules["torchaudio"] = _ta
sys.modules["torchaudio.functional"] = _taf

_x = torch.randn(40, dtype=torch.float64)
_h = torch.randn(7, dtype=torch.float64)
assert np.allclose(_convolve(_x, _h).numpy(), np.convolve(_x.numpy(), _h.numpy(), "valid"), rtol=0, atol=1e-14)

from lib.real_equalization import SecondVolterraV2VAE, LinearV2VAE, SecondVolterraV2VAENoNoiseScaling  # noqa: E402
from lib.utility import calc_ser_from_probs  # noqa: E402

from volterra_vae_amd.synth import nonlinear_isi, pam_constellation  # noqa: E402


def randomise(model, rng, with_k2=True):
    """Perturb every parameter so that no term of the ELBO is identically zero."""
    L = model.channel_memory
    with torch.no_grad():
        model.channel_h1 += torch.from_numpy(0.1 * rng.standard_normal(L)).to(model.dtype)
        model.channel_h2 += torch.from_numpy(0.05 * rng.standard_normal((L, L))).to(model.dtype)
        model.noise_scaling *= torch.from_numpy(1.0 + 0.2 * rng.random(len(model.constellation))).to(model.dtype)
        if with_k2:
            eq = model.equaliser
            eq.kernel1 += torch.from_numpy(0.1 * rng.standard_normal(eq.kernel1.shape)).to(model.dtype)
            eq.kernel2 += torch.from_numpy(0.02 * rng.standard_normal(eq.kernel2.shape)).to(model.dtype)
        else:
            w = model.equaliser.weight
            w += torch.from_numpy(0.1 * rng.standard_normal(tuple(w.shape))).to(model.dtype)


def single_step_case(name, L, S, l1, l2, B, order, dtype, seed, linear_eq=False, no_scaling=False):
    rng = np.random.default_rng(seed)
    const = pam_constellation(order)
    rx, syms = nonlinear_isi(B, const, S, 16.0, rng)
    nv0 = 0.05
    kw = dict(channel_memory=L, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
              dtype=dtype, noise_variance=nv0, adaptive_noise_variance=True)
    if linear_eq:
        model = LinearV2VAE(equaliser_n_taps=l1, **kw)
    elif no_scaling:
        model = SecondVolterraV2VAENoNoiseScaling(equaliser_n_lags1=l1, equaliser_n_lags2=l2, **kw)
    else:
        model = SecondVolterraV2VAE(equaliser_n_lags1=l1, equaliser_n_lags2=l2, **kw)
    randomise(model, rng, with_k2=not linear_eq)
    out = dict(L=L, S=S, l1=l1, l2=l2, B=B, order=order, linear_eq=int(linear_eq), no_scaling=int(no_scaling),
               rx=rx, constellation=const, noise_variance_in=nv0,
               h1=model.channel_h1.detach().numpy().copy(), h2=model.channel_h2.detach().numpy().copy(),
               noise_scaling=model.noise_scaling.detach().numpy().copy())
    if linear_eq:
        out["eq_w"] = model.equaliser.weight.detach().numpy().copy()
    else:
        out["k1"] = model.equaliser.kernel1.detach().numpy().copy()
        out["k2"] = model.equaliser.kernel2.detach().numpy().copy()
    y = torch.from_numpy(rx).type(dtype)
    xhat = model.forward(y)
    q = model._soft_demapping(xhat)
    loss = model._calculate_loss(xhat, y)
    loss.backward()
    W = len(rx) - L + 1
    out.update(xhat=xhat.detach().numpy(), q=q.detach().numpy(), loss=loss.detach().numpy(),
               E=(model.noise_variance.detach().numpy() * W), noise_variance_out=model.noise_variance.detach().numpy(),
               g_h1=model.channel_h1.grad.numpy(), g_h2=model.channel_h2.grad.numpy())
    if not no_scaling:
        out["g_noise_scaling"] = model.noise_scaling.grad.numpy()
    if linear_eq:
        out["g_eq_w"] = model.equaliser.weight.grad.numpy()
    else:
        out["g_k1"] = model.equaliser.kernel1.grad.numpy()
        out["g_k2"] = model.equaliser.kernel2.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"v2vae_real_step_{name}.npz"), **out)
    print(f"step case {name}: loss={float(loss):.6f}")


def trajectory_case(name, L, S, l1, l2, B, n_batches, order, dtype, seed, lr, lr_schedule, linear_eq=False,
                    store_rx=True, n_val=2000):
    rng = np.random.default_rng(seed)
    const = pam_constellation(order)
    rx, syms = nonlinear_isi(B * n_batches, const, S, 16.0, rng)
    rx_val, syms_val = nonlinear_isi(n_val, const, S, 16.0, rng)
    nv0 = 10 ** (-16.0 / 10)
    kw = dict(channel_memory=L, learning_rate=lr, constellation=const, samples_per_symbol=S, batch_size=B,
              dtype=dtype, noise_variance=nv0, adaptive_noise_variance=True, lr_schedule=lr_schedule)
    if linear_eq:
        model = LinearV2VAE(equaliser_n_taps=l1, **kw)
    else:
        model = SecondVolterraV2VAE(equaliser_n_lags1=l1, equaliser_n_lags2=l2, **kw)
    model.initialize_optimizer()
    nv_traj = []
    orig = model._update_model

    def spy(loss):
        nv_traj.append(float(model.noise_variance))
        orig(loss)
    model._update_model = spy
    y_eq, loss_curve = model.fit(rx)
    q_val = model.estimate_symbol_probs(rx_val)
    y_apply = model.apply(rx_val)
    ser, delay = calc_ser_from_probs(q_val, syms_val, const)
    out = dict(L=L, S=S, l1=l1, l2=l2, B=B, n_batches=n_batches, order=order, lr=lr,
               lr_schedule=(lr_schedule or "none"), linear_eq=int(linear_eq), seed=seed, n_val=n_val,
               rx_checksum=float(np.sum(rx * np.cos(np.arange(len(rx))))),
               constellation=const, noise_variance_in=nv0, loss_curve=loss_curve, noise_variance_traj=np.array(nv_traj),
               h1=model.channel_h1.detach().numpy(), h2=model.channel_h2.detach().numpy(),
               noise_scaling=model.noise_scaling.detach().numpy(),
               ser=ser, ser_delay=delay)
    if store_rx:
        out.update(rx=rx, rx_val=rx_val, syms_val=syms_val, y_eq=y_eq, q_val=q_val.astype(np.float32), y_apply=y_apply)
    else:
        out.update(decisions=np.argmax(q_val, axis=0).astype(np.int8))
    if linear_eq:
        out["eq_w"] = model.equaliser.weight.detach().numpy()
    else:
        out["k1"] = model.equaliser.kernel1.detach().numpy()
        out["k2"] = model.equaliser.kernel2.detach().numpy()
    np.savez_compressed(os.path.join(HERE, f"v2vae_real_fit_{name}.npz"), **out)
    print(f"fit case {name}: loss[0]={loss_curve[0]:.4f} loss[-1]={loss_curve[-1]:.4f} ser={ser:.4e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        single_step_case(f"cfg1_{tag}", 15, 2, 25, 25, 500, 4, dt, 11)
        single_step_case(f"L7S1_{tag}", 7, 1, 5, 3, 64, 4, dt, 12)
        single_step_case(f"L8S2_{tag}", 8, 2, 9, 7, 96, 4, dt, 13)
        single_step_case(f"L9S3_{tag}", 9, 3, 7, 5, 60, 8, dt, 14)
        single_step_case(f"lin_L15S2_{tag}", 15, 2, 25, 0, 200, 4, dt, 15, linear_eq=True)
        single_step_case(f"noscale_L15S2_{tag}", 15, 2, 25, 25, 200, 4, dt, 16, no_scaling=True)
        trajectory_case(f"cfg1_{tag}", 15, 2, 25, 25, 500, 20, 4, dt, 21, 5e-3, "step")
        trajectory_case(f"cfg3_{tag}", 15, 2, 25, 15, 500, 20, 4, dt, 22, 5e-4, None)
        trajectory_case(f"lin_{tag}", 15, 2, 25, 0, 300, 20, 4, dt, 23, 5e-3, "step", linear_eq=True)
    # long fit (inputs regenerated from the seed by the test; rx_checksum guards the generator)
    trajectory_case("long_cfg1_f32", 15, 2, 25, 25, 500, 400, 4, torch.float32, 31, 5e-3, "step",
                    store_rx=False, n_val=20000)
