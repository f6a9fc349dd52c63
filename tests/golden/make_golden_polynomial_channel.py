"""Golden vectors of the reference's memory-polynomial and Hammerstein channel VAEs (lib/real_equalization.py:
SecondVolterraMPVAE :1387-1410 with VAEMemoryPolynomialForward :1224-1384, SecondVolterraHVAE :1202-1221 with
VAEHammersteinForward :1040-1199), generated like make_golden.py (same torchaudio stand-in, asserted there).

Output: tests/golden/poly_real_{step,fit}_*.npz (inputs + expected outputs only; no reference code).
Usage:  python tests/golden/make_golden_polynomial_channel.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402,F401  (registers the torchaudio stand-in and the import paths)

from lib.real_equalization import SecondVolterraHVAE, SecondVolterraMPVAE  # noqa: E402
from lib.utility import calc_ser_from_probs  # noqa: E402

from volterra_vae_amd.synth import nonlinear_isi, pam_constellation  # noqa: E402


def build(kind, L, S, l1, l2, B, const, dtype, nv0, lr, lr_schedule, noise_scaling=True):
    kw = dict(channel_memory=L, learning_rate=lr, constellation=const, samples_per_symbol=S, batch_size=B, dtype=dtype,
              noise_variance=nv0, adaptive_noise_variance=True, lr_schedule=lr_schedule, equaliser_n_lags1=l1,
              equaliser_n_lags2=l2)
    if kind == "mpvae":
        return SecondVolterraMPVAE(noise_scaling=noise_scaling, **kw)
    return SecondVolterraHVAE(**kw)


def channel_params(model, kind):
    if kind == "mpvae":
        return dict(h1=model.channel_h1.detach().numpy().copy(), h2=model.channel_h2.detach().numpy().copy())
    return dict(h=model.channel_h.detach().numpy().copy(), alpha1=model.alpha_1.detach().numpy().copy(),
                alpha2=model.alpha_2.detach().numpy().copy())


def step_case(name, kind, L, S, l1, l2, B, order, dtype, seed):
    rng = np.random.default_rng(seed)
    const = pam_constellation(order)
    rx, _ = nonlinear_isi(B, const, S, 16.0, rng)
    model = build(kind, L, S, l1, l2, B, const, dtype, 0.05, 5e-3, "step")
    with torch.no_grad():
        if kind == "mpvae":
            model.channel_h1 += torch.from_numpy(0.1 * rng.standard_normal(L)).to(dtype)
            model.channel_h2 += torch.from_numpy(0.03 * rng.standard_normal(L)).to(dtype)
        else:
            model.channel_h += torch.from_numpy(0.1 * rng.standard_normal(L)).to(dtype)
            model.alpha_1 += 0.07
            model.alpha_2 += 0.05
        model.equaliser.kernel1 += torch.from_numpy(0.1 * rng.standard_normal(model.equaliser.kernel1.shape)).to(dtype)
        model.equaliser.kernel2 += torch.from_numpy(0.02 * rng.standard_normal(model.equaliser.kernel2.shape)).to(dtype)
        model.noise_scaling += torch.from_numpy(0.1 * rng.random(len(const))).to(dtype)
    out = dict(kind=kind, L=L, S=S, l1=l1, l2=l2, B=B, order=order, rx=rx, constellation=const, noise_variance_in=0.05,
               k1=model.equaliser.kernel1.detach().numpy().copy(), k2=model.equaliser.kernel2.detach().numpy().copy(),
               noise_scaling=model.noise_scaling.detach().numpy().copy(), **channel_params(model, kind))
    y = torch.from_numpy(rx).type(dtype)
    xhat = model.forward(y)
    q = model._soft_demapping(xhat)
    loss = model._calculate_loss(xhat, y)
    loss.backward()
    out.update(xhat=xhat.detach().numpy(), q=q.detach().numpy(), loss=loss.detach().numpy(),
               noise_variance_out=model.noise_variance.detach().numpy(), g_k1=model.equaliser.kernel1.grad.numpy(),
               g_k2=model.equaliser.kernel2.grad.numpy(), g_noise_scaling=model.noise_scaling.grad.numpy())
    if kind == "mpvae":
        out.update(g_h1=model.channel_h1.grad.numpy(), g_h2=model.channel_h2.grad.numpy())
    else:
        out.update(g_h=model.channel_h.grad.numpy(), g_alpha1=model.alpha_1.grad.numpy(), g_alpha2=model.alpha_2.grad.numpy())
    np.savez_compressed(os.path.join(HERE, f"poly_real_step_{name}.npz"), **out)
    print(f"step case {name}: loss={float(loss):.6f}")


def fit_case(name, kind, L, S, l1, l2, B, n_batches, order, dtype, seed, lr, lr_schedule, noise_scaling=True, n_val=2000):
    rng = np.random.default_rng(seed)
    const = pam_constellation(order)
    rx, _ = nonlinear_isi(B * n_batches, const, S, 16.0, rng)
    rx_val, syms_val = nonlinear_isi(n_val, const, S, 16.0, rng)
    nv0 = 10 ** (-16.0 / 10)
    model = build(kind, L, S, l1, l2, B, const, dtype, nv0, lr, lr_schedule, noise_scaling)
    model.initialize_optimizer()
    y_eq, loss_curve = model.fit(rx)
    q_val = model.estimate_symbol_probs(rx_val)
    ser, delay = calc_ser_from_probs(q_val, syms_val, const)
    out = dict(kind=kind, L=L, S=S, l1=l1, l2=l2, B=B, n_batches=n_batches, order=order, lr=lr, lr_schedule=(lr_schedule or "none"),
               learn_noise_scaling=int(noise_scaling), constellation=const, noise_variance_in=nv0, rx=rx, rx_val=rx_val,
               syms_val=syms_val, y_eq=y_eq, loss_curve=loss_curve, q_val=q_val.astype(np.float32),
               noise_variance_out=float(model.noise_variance), k1=model.equaliser.kernel1.detach().numpy(),
               k2=model.equaliser.kernel2.detach().numpy(), noise_scaling=model.noise_scaling.detach().numpy(), ser=ser,
               ser_delay=delay, **channel_params(model, kind))
    np.savez_compressed(os.path.join(HERE, f"poly_real_fit_{name}.npz"), **out)
    print(f"fit case {name}: loss[0]={loss_curve[0]:.4f} loss[-1]={loss_curve[-1]:.4f} ser={ser:.4e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        step_case(f"mpvae_{tag}", "mpvae", 15, 2, 25, 25, 200, 4, dt, 61)
        step_case(f"mpvae_L8S1_{tag}", "mpvae", 8, 1, 7, 5, 80, 8, dt, 62)
        step_case(f"mpvae_L10S3_{tag}", "mpvae", 10, 3, 9, 7, 90, 4, dt, 63)
        step_case(f"hvae_{tag}", "hvae", 15, 2, 25, 25, 200, 4, dt, 64)
        step_case(f"hvae_L8S1_{tag}", "hvae", 8, 1, 7, 5, 80, 8, dt, 65)
        fit_case(f"mpvae_{tag}", "mpvae", 15, 2, 25, 25, 500, 20, 4, dt, 71, 5e-3, "step")
        fit_case(f"mpvae_fixed_scaling_{tag}", "mpvae", 15, 2, 25, 25, 300, 20, 4, dt, 72, 5e-3, None, noise_scaling=False)
        fit_case(f"hvae_{tag}", "hvae", 15, 2, 25, 25, 500, 20, 4, dt, 73, 5e-3, "step")
