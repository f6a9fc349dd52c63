"""Monte-Carlo check of the closed-form ELBO expectations of the complex V2VAE (no reference implementation
exists for it).  In the manner of the reference's derivation scripts (test_volterra_expectations.py:139-367:
draw many symbol sequences, evaluate the channel model on each, compare the empirical mean of the summed squared
error with the closed form and require it to lie within a confidence band), here for the holomorphic
second-order complex channel with independent per-axis categorical posteriors."""
import numpy as np
import torch

from oracle import v2vae_oracle as orc


def test_closed_form_E_matches_monte_carlo():
    rng = np.random.default_rng(123)
    L, S, B, M = 5, 2, 12, 4
    N = B * S
    lv = np.arange(-(M - 1), M, 2).astype(float)
    levels = np.stack((lv, lv))
    # arbitrary per-symbol posteriors: obtained from random xhat through the soft demapper
    xhat = torch.from_numpy(rng.uniform(-3, 3, size=(2, B)))
    nv = torch.tensor(0.8, dtype=torch.float64)
    scaling = torch.ones(2, M, dtype=torch.float64)
    prior = torch.full((2, M), 1.0 / M, dtype=torch.float64)
    h1 = torch.from_numpy(0.4 * rng.standard_normal((2, L)))
    h2 = torch.from_numpy(0.15 * rng.standard_normal((2, L, L)))
    y = torch.from_numpy(rng.standard_normal((2, N)))
    loss, E, kl = orc.complex_v2vae_loss(xhat, y, h1[0], h1[1], h2[0], h2[1], torch.from_numpy(levels), prior, nv, scaling, S)
    E = float(E)

    q = np.stack([orc.soft_demap(xhat[c], torch.from_numpy(lv), nv, scaling[c]).numpy() for c in range(2)])  # [2, M, B]
    hc = (h1[0] + 1j * h1[1]).numpy()
    Hc = ((h2[0] + h2[0].T) + 1j * (h2[1] + h2[1].T)).numpy()
    yc = (y[0] + 1j * y[1]).numpy()
    pre = (L - 1) // 2
    W = N - L + 1
    n_draws = 200000
    cdf = np.cumsum(q, axis=1)                                    # [2, M, B]
    u = rng.random((n_draws, 2, 1, B))
    idx = (u > cdf[None]).sum(axis=2).clip(0, M - 1)              # [n, 2, B]
    sym = lv[idx[:, 0]] + 1j * lv[idx[:, 1]]                      # [n, B]
    x = np.zeros((n_draws, N), dtype=complex)
    x[:, ::S] = sym
    tot = np.zeros(n_draws)
    for n in range(W):
        xw = x[:, n:n + L][:, ::-1]                               # tap i reads sample n + L - 1 - i
        f = xw @ hc + np.einsum("ni,ij,nj->n", xw, Hc, xw)
        tot += np.abs(yc[pre + n] - f) ** 2
    mean, sem = tot.mean(), tot.std(ddof=1) / np.sqrt(n_draws)
    assert abs(mean - E) < 4 * sem, f"closed form {E:.4f} vs Monte-Carlo {mean:.4f} +- {sem:.4f}"


def test_complex_model_reduces_to_real_model_for_real_data():
    """Im = 0 everywhere (data, kernels; the Q axis has the single level 0): the complex loss's I part is the real loss."""
    rng = np.random.default_rng(5)
    L, S, B = 7, 2, 40
    lv = torch.tensor([-3.0, -1.0, 1.0, 3.0], dtype=torch.float64)
    xr = torch.from_numpy(rng.uniform(-3, 3, B))
    y = torch.from_numpy(rng.standard_normal(B * S))
    h1 = torch.from_numpy(0.3 * rng.standard_normal(L)); h1[L // 2] += 1
    h2 = torch.from_numpy(0.1 * rng.standard_normal((L, L)))
    nv = torch.tensor(0.5, dtype=torch.float64)
    sc = torch.ones(4, dtype=torch.float64)
    pr = torch.full((4,), 0.25, dtype=torch.float64)
    H = h2 + h2.T
    loss_r, E_r, kl_r = orc.elbo_window_form(xr[None], y[None], h1[None], H[None], torch.diagonal(H)[None], lv[None],
                                             pr[None], nv, sc[None], S)
    z = torch.zeros
    levels_c = torch.stack((lv, torch.zeros(4, dtype=torch.float64)))   # Q axis: all levels 0 -> x_Q = 0 surely
    loss_c, E_c, kl_c = orc.complex_v2vae_loss(torch.stack((xr, z(B, dtype=torch.float64))), torch.stack((y, z(B * S, dtype=torch.float64))),
                                               h1, z(L, dtype=torch.float64), h2, z(L, L, dtype=torch.float64), levels_c,
                                               torch.stack((pr, pr)), nv, torch.stack((sc, sc)), S)
    assert abs(float(E_c) - float(E_r)) < 1e-10 * abs(float(E_r))


def test_complex_model_reduces_to_reference_linear_vae_loss():
    """With a zero second-order kernel and unit noise scaling the complex V2VAE loss is the loss of the reference's
    complex LinearVAE (lib/complex_equalization.py:819-861), whose channel filter is a cross-correlation
    (Conv1d) and whose conjugation convention makes the effective tap w_re - j w_im acting on x... the equivalent
    true-convolution taps are the flipped ones."""
    rng = np.random.default_rng(9)
    L, S, B, M = 7, 2, 30, 4
    N = B * S
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float64)
    levels = torch.stack((lv, lv))
    prior = torch.full((2, M), 1.0 / M, dtype=torch.float64)
    xhat = torch.from_numpy(rng.uniform(-3, 3, (2, B)))
    y = torch.from_numpy(rng.standard_normal((2, N)))
    w = torch.from_numpy(0.4 * rng.standard_normal((2, L)))  # reference encoder weight [1, 2, L] -> rows (w0, w1)
    nv = torch.tensor(0.7, dtype=torch.float64)
    # reference arithmetic, restated from lib/complex_equalization.py:819-861
    pre, end = (L - 1) // 2, -(L - 1) // 2
    q = torch.cat([orc.soft_demap(xhat[c], lv, nv, torch.ones(M, dtype=torch.float64)) for c in range(2)])  # [2M, B]
    qs = q[:, pre:end]
    kl = torch.sum(qs * torch.log(qs / (1.0 / M) + 1e-12))
    amp = torch.cat((lv, lv))
    ex = torch.zeros(2, N, dtype=torch.float64); ex2 = torch.zeros(2, N, dtype=torch.float64)
    ex[0, ::S] = (q[:M] * lv[:, None]).sum(0); ex[1, ::S] = (q[M:] * lv[:, None]).sum(0)
    ex2[0, ::S] = (q[:M] * lv[:, None] ** 2).sum(0); ex2[1, ::S] = (q[M:] * lv[:, None] ** 2).sum(0)
    conv = lambda inp: torch.nn.functional.conv1d(inp[None], w[None])[0, 0]
    d_real = conv(ex * torch.tensor([[1.0], [-1.0]], dtype=torch.float64))
    d_imag = conv(torch.roll(ex, 1, dims=0))
    hsq = (w ** 2).sum(0)
    var = (ex2.sum(0) - (ex ** 2).sum(0))
    hsqconv = torch.nn.functional.conv1d(var[None, None], hsq[None, None])[0, 0]
    e_hx2 = torch.sum(d_real ** 2 + d_imag ** 2 + hsqconv)
    E_ref = torch.sum(y[:, pre:end] ** 2) - 2 * (y[0, pre:end] @ d_real) - 2 * (y[1, pre:end] @ d_imag) + e_hx2
    loss_ref = kl + (N - L + 1) * torch.log(E_ref)
    # our model: true convolution with taps h[i] = (w0 + j w1)[L - 1 - i]
    h1r, h1i = torch.flip(w[0], (0,)), torch.flip(w[1], (0,))
    zero = torch.zeros(L, L, dtype=torch.float64)
    loss, E, kl2 = orc.complex_v2vae_loss(xhat, y, h1r, h1i, zero, zero, levels, prior, nv, torch.ones(2, M, dtype=torch.float64), S)
    assert abs(float(E) - float(E_ref)) < 1e-10 * abs(float(E_ref))
    assert abs(float(loss) - float(loss_ref)) < 1e-10 * abs(float(loss_ref))
