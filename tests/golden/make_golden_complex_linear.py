"""Golden vectors of the reference's COMPLEX linear-channel VAE (lib/complex_equalization.py: LinearVAE :890-916 on
VAELinearForward :746-887), imported unmodified (this module of the reference needs no stand-in).

They pin the complex kernels (matrix-core ELBO and equaliser included) against the reference on the subset of the complex
model that the reference implements: linear channel, linear equaliser.  For the float64 run torch's default dtype is
switched to float64, because the reference's forward() allocates its output with torch.empty() (:910).

Output: tests/golden/vae_complex_{step,fit}_*.npz (inputs + expected outputs only).
Usage:  python tests/golden/make_golden_complex_linear.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from lib.complex_equalization import LinearVAE  # noqa: E402

from volterra_vae_amd.synth import nonlinear_isi, qam_constellation  # noqa: E402


def build(L, S, l1, B, const, dtype, nv0, lr):
    return LinearVAE(encoder_n_taps=L, learning_rate=lr, constellation=const, samples_per_symbol=S, batch_size=B, dtype=dtype,
                     noise_variance=nv0, adaptive_noise_variance=True, decoder_n_taps=l1)


def step_case(name, L, S, l1, B, order, dtype, seed):
    rng = np.random.default_rng(seed)
    const = qam_constellation(order)
    rx, _ = nonlinear_isi(B, const, S, 16.0, rng)
    m = build(L, S, l1, B, const, dtype, 0.05, 5e-3)
    with torch.no_grad():
        m.encoder.weight += torch.from_numpy(0.1 * rng.standard_normal(tuple(m.encoder.weight.shape))).to(dtype)
        m.decoder.weight += torch.from_numpy(0.1 * rng.standard_normal(tuple(m.decoder.weight.shape))).to(dtype)
    out = dict(L=L, S=S, l1=l1, B=B, order=order, rx=rx, constellation=const, noise_variance_in=0.05,
               enc_w=m.encoder.weight.detach().numpy().copy(), dec_w=m.decoder.weight.detach().numpy().copy())
    y = torch.from_numpy(m.to_real_stacked(rx)).type(dtype)
    xhat = m.forward(y)
    q = m._soft_demapping(xhat)
    loss = m._calculate_loss(xhat, y)
    loss.backward()
    out.update(xhat=xhat.detach().numpy(), q=q.detach().numpy(), loss=loss.detach().numpy(),
               noise_variance_out=m.noise_variance.detach().numpy(), g_enc_w=m.encoder.weight.grad.numpy(),
               g_dec_w=m.decoder.weight.grad.numpy())
    np.savez_compressed(os.path.join(HERE, f"vae_complex_step_{name}.npz"), **out)
    print(f"step case {name}: loss={float(loss):.6f}")


def fit_case(name, L, S, l1, B, n_batches, order, dtype, seed, lr, n_val=2000):
    rng = np.random.default_rng(seed)
    const = qam_constellation(order)
    rx, _ = nonlinear_isi(B * n_batches, const, S, 16.0, rng)
    rx_val, _ = nonlinear_isi(n_val, const, S, 16.0, rng)
    nv0 = 10 ** (-16.0 / 10)
    m = build(L, S, l1, B, const, dtype, nv0, lr)
    y_eq = m.fit(rx)
    y_apply = m.apply(rx_val)
    out = dict(L=L, S=S, l1=l1, B=B, n_batches=n_batches, order=order, lr=lr, constellation=const, noise_variance_in=nv0,
               rx=rx, rx_val=rx_val, y_eq=y_eq, y_apply=y_apply, noise_variance_out=float(m.noise_variance),
               enc_w=m.encoder.weight.detach().numpy(), dec_w=m.decoder.weight.detach().numpy())
    np.savez_compressed(os.path.join(HERE, f"vae_complex_fit_{name}.npz"), **out)
    print(f"fit case {name}: noise variance {float(m.noise_variance):.6f}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        torch.set_default_dtype(dt)
        step_case(f"L15S2_{tag}", 15, 2, 25, 200, 16, dt, 61)
        step_case(f"L25S2_{tag}", 25, 2, 25, 300, 16, dt, 62)     # the matrix-core shape of the headline benchmark
        step_case(f"L9S1_{tag}", 9, 1, 7, 120, 4, dt, 63)
        fit_case(f"L25S2_{tag}", 25, 2, 25, 400, 20, 16, dt, 71, 2e-3)
    torch.set_default_dtype(torch.float32)
