"""GPU tests of the persistent fit kernel (one workgroup per model, the whole mini-batch loop in one launch).

The persistent kernel runs the same stage code in the same summation order as the per-batch kernel sequence on
the generic kernels, so for the real model the two routes must agree bit for bit; the reference-pinned route is
the per-batch one (tests/test_gpu_dropin.py), which pins this one through that equality."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _model(g, persistent):
    from test_gpu_dropin import make_model
    m = make_model(g)
    m.persistent_fit = persistent
    m.initialize_optimizer()
    return m


@pytest.mark.parametrize("case", ["cfg1", "cfg3", "lin"])
def test_persistent_fit_is_bitwise_the_per_batch_fit(case, monkeypatch):
    monkeypatch.setenv("V2VAE_FORCE_GENERIC", "1")  # the stage-by-stage persistent kernel, not the on-chip one
    g = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{case}_f32.npz"))
    a, b = _model(g, True), _model(g, False)
    ya, la = a.fit(g["rx"])
    yb, lb = b.fit(g["rx"])
    assert a._persistent(int(g["B"])) is not None, "shape did not take the persistent route"
    assert np.array_equal(ya, yb)
    assert np.array_equal(la, lb)
    assert torch.equal(a._pack, b._pack)
    assert torch.equal(a._noise_var, b._noise_var)
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg)
    assert torch.equal(a.optimizer.max_exp_avg_sq, b.optimizer.max_exp_avg_sq)
    assert a.optimizer.step_count == b.optimizer.step_count
    # a second fit() continues the optimiser state identically
    ya, la = a.fit(g["rx"])
    yb, lb = b.fit(g["rx"])
    assert np.array_equal(ya, yb) and np.array_equal(la, lb) and torch.equal(a._pack, b._pack)


def _rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64).reshape(-1)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64).reshape(-1)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("case", ["cfg1", "cfg3", "lin"])
def test_on_chip_fit_matches_per_batch_fit_and_reference(case):
    """Small real shapes run the whole fit inside one workgroup (different summation order, same mathematics):
    20 dependent Adam steps must stay within fp32 rounding of the per-batch kernels and of the reference run."""
    g = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{case}_f32.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{case}_f64.npz"))
    a, b = _model(g, True), _model(g, False)
    ya, la = a.fit(g["rx"])
    yb, lb = b.fit(g["rx"])
    assert not np.array_equal(ya, yb) or case == "never", "expected the on-chip kernel, got the bitwise-equal generic one"
    for name, x, y in (("y_eq", ya, yb), ("loss", la, lb), ("h1", a.channel_h1, b.channel_h1),
                       ("scaling", a.noise_scaling, b.noise_scaling), ("nv", a._noise_var, b._noise_var)):
        assert _rel(x, y) < 2e-5, (name, _rel(x, y))
    assert _rel(a.channel_h2, b.channel_h2) < 2e-4
    assert _rel(a._pack, b._pack) < 5e-5
    # against the reference's own fp64 run, with the reference's fp32-vs-fp64 spread as the yardstick
    for key, ours, tol in (("loss_curve", la, 1e-5), ("y_eq", ya, 1e-5), ("h1", a.channel_h1, 1e-5), ("h2", a.channel_h2, 1e-4)):
        own = _rel(g[key], g64[key])
        assert _rel(ours, g64[key]) < tol + own, (key, _rel(ours, g64[key]), own)


def test_on_chip_fit_long_batches_and_phases():
    """Batches longer than one window chunk, three samples per symbol, even channel length."""
    from volterra_vae_amd.real_equalization import SecondVolterraV2VAE
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    const = pam_constellation(4)
    for S, B, L, l1, l2 in ((3, 700, 12, 13, 9), (1, 300, 8, 7, 7), (2, 1100, 14, 31, 31)):
        rx, _ = nonlinear_isi(12 * B, const, S, 17.0, np.random.default_rng(S))
        ms = []
        for persistent in (True, False):
            m = SecondVolterraV2VAE(channel_memory=L, learning_rate=3e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                                    noise_variance=0.1, torch_device=torch.device("cuda"), equaliser_n_lags1=l1,
                                    equaliser_n_lags2=l2)
            m.persistent_fit = persistent
            m.initialize_optimizer()
            ms.append((m,) + m.fit(rx))
        (a, ya, la), (b, yb, lb) = ms
        assert _rel(ya, yb) < 2e-5 and _rel(la, lb) < 2e-5, (S, B, _rel(ya, yb), _rel(la, lb))
        assert _rel(a._pack, b._pack) < 1e-4, (S, B, _rel(a._pack, b._pack))


def test_sweep_of_models_equals_individual_fits():
    """Three models with their own sequences and start noise variances in one launch == three separate fits."""
    from volterra_vae_amd.ops import PersistentFit
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    dev = torch.device("cuda")
    S, B, L, l1, l2, C, nb, M = 2, 200, 9, 11, 7, 4, 20, 3
    const = pam_constellation(C)
    rng = np.random.default_rng(5)
    rxs = np.stack([nonlinear_isi(nb * B, const, S, snr, rng)[0] for snr in (12.0, 16.0, 20.0)]).astype(np.float32)
    y = torch.from_numpy(rxs).view(M, 1, -1).to(dev).contiguous()
    lv = torch.tensor(const, dtype=torch.float32, device=dev).view(1, -1)
    prior = torch.full_like(lv, 1.0 / C)
    lrs = [5e-3 * (0.8 ** (n // 5)) for n in range(nb)]

    def fresh(m):
        pf = PersistentFit(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=C, batch_symbols=B, n_models=m, device=dev)
        n = pf.sizes
        prm = torch.zeros((m, pf.pack), dtype=torch.float32, device=dev)
        prm[:, L // 2] = 1.0
        prm[:, n[0] + n[1] + l1 // 2] = 1.0
        prm[:, n[0] + n[1] + n[2] + n[3]:] = 1.0
        st = [torch.zeros_like(prm) for _ in range(4)]
        return pf, prm, st

    pf, prm, st = fresh(M)
    nv = torch.tensor([0.2, 0.1, 0.05], dtype=torch.float32, device=dev)
    yq = torch.zeros((M, 1, nb * B), dtype=torch.float32, device=dev)
    lc = torch.zeros((M, nb), dtype=torch.float32, device=dev)
    pf.run(y=y, levels=lv, prior=prior, params=prm, grads=st[0], adam_m=st[1], adam_v=st[2], adam_vmax=st[3], noise_var=nv,
           lrs=lrs, y_eq=yq, loss_curve=lc)
    torch.cuda.synchronize()
    assert torch.isfinite(prm).all() and torch.isfinite(lc).all()
    for m in range(M):
        pf1, prm1, st1 = fresh(1)
        nv1 = torch.tensor([[0.2, 0.1, 0.05][m]], dtype=torch.float32, device=dev)
        yq1 = torch.zeros((1, 1, nb * B), dtype=torch.float32, device=dev)
        lc1 = torch.zeros((1, nb), dtype=torch.float32, device=dev)
        pf1.run(y=y[m].contiguous(), levels=lv, prior=prior, params=prm1, grads=st1[0], adam_m=st1[1], adam_v=st1[2],
                adam_vmax=st1[3], noise_var=nv1, lrs=lrs, y_eq=yq1, loss_curve=lc1)
        torch.cuda.synchronize()
        assert torch.equal(prm1[0], prm[m]) and torch.equal(yq1[0], yq[m]) and torch.equal(lc1[0], lc[m])
        assert torch.equal(nv1[0], nv[m])
    # the three models saw different data: they must have diverged
    assert not torch.equal(prm[0], prm[1])


def test_persistent_complex_matches_per_batch_steps():
    """Complex model through the persistent kernel (generic stage code) vs per-batch steps on the matrix-core
    kernels: same mathematics, different summation order -> agreement to fp32 rounding over a short trajectory."""
    from volterra_vae_amd.ops import PersistentFit, Plan
    from volterra_vae_amd.synth import nonlinear_isi, qam_constellation
    dev = torch.device("cuda")
    S, B, L, l1, l2, M, nb = 2, 256, 7, 9, 5, 4, 6
    rx, _ = nonlinear_isi(nb * B, qam_constellation(M * M), S, 18.0, np.random.default_rng(11))
    y = torch.from_numpy(np.stack((rx.real, rx.imag))).float().to(dev).contiguous()
    lvl = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev)
    lv = torch.stack((lvl, lvl)).contiguous()
    prior = torch.full_like(lv, 1.0 / M)
    lrs = [2e-3] * nb

    pf = PersistentFit(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, batch_symbols=B, n_models=1, device=dev)
    n = pf.sizes

    def init():
        prm = torch.zeros(pf.pack, dtype=torch.float32, device=dev)
        prm[L // 2] = 1.0
        prm[n[0] + n[1] + l1 // 2] = 1.0
        prm[n[0] + n[1] + n[2] + n[3]:] = 1.0
        return prm, [torch.zeros_like(prm) for _ in range(4)], torch.tensor([0.1], dtype=torch.float32, device=dev)

    prm, st, nv = init()
    yq = torch.zeros((1, 2, nb * B), dtype=torch.float32, device=dev)
    lc = torch.zeros((1, nb), dtype=torch.float32, device=dev)
    pf.run(y=y, levels=lv, prior=prior, params=prm.view(1, -1), grads=st[0].view(1, -1), adam_m=st[1].view(1, -1),
           adam_v=st[2].view(1, -1), adam_vmax=st[3].view(1, -1), noise_var=nv, lrs=lrs, y_eq=yq, loss_curve=lc)

    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device=dev)
    prm2, st2, nv2 = init()
    xh = torch.zeros((2, B), dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    for i in range(nb):
        yb = y[:, i * B * S:(i + 1) * B * S].contiguous()
        plan.train_step(y=yb, levels=lv, prior=prior, params=prm2, grads=st2[0], adam_m=st2[1], adam_v=st2[2], adam_vmax=st2[3],
                        noise_var=nv2, xhat=xh, loss_out=loss, lr=lrs[i], step=i + 1)
        assert torch.allclose(xh, yq[0, :, i * B:(i + 1) * B], rtol=1e-4, atol=1e-5)
        assert abs(float(loss) - float(lc[0, i])) < 1e-4 * abs(float(loss))
    err = float(torch.linalg.norm(prm - prm2) / torch.linalg.norm(prm2))
    assert err < 1e-4, err


def test_persistent_rejects_bad_arguments():
    from volterra_vae_amd._lib import V2VAEError
    from volterra_vae_amd.ops import PersistentFit
    dev = torch.device("cuda")
    pf = PersistentFit(ns=1, sps=2, l1=11, pad1=5, l2=7, L=9, C_levels=4, batch_symbols=100, n_models=2, device=dev)
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        pf.run(y=z(1, 100), levels=z(1, 4), prior=z(1, 4), params=z(2, pf.pack), grads=z(2, pf.pack), adam_m=z(2, pf.pack),
               adam_v=z(2, pf.pack), adam_vmax=z(2, pf.pack), noise_var=z(2), lrs=[1e-3] * 3)  # sequence too short
    with pytest.raises(V2VAEError):
        PersistentFit(ns=1, sps=2, l1=11, pad1=5, l2=8, L=9, C_levels=4, batch_symbols=100, n_models=1, device=dev)  # even lags


def test_sweep_class_equals_individual_models():
    """SecondVolterraV2VAESweep (one launch for all models) == one drop-in SecondVolterraV2VAE per model."""
    from volterra_vae_amd.real_equalization import SecondVolterraV2VAE
    from volterra_vae_amd.sweep import SecondVolterraV2VAESweep
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    S, B, L, l1, l2, nb, M = 2, 250, 15, 25, 25, 20, 4
    const = pam_constellation(4)
    rng = np.random.default_rng(9)
    snrs = (10.0, 14.0, 18.0, 22.0)
    rxs = np.stack([nonlinear_isi(nb * B, const, S, snr, rng)[0] for snr in snrs])
    nv0 = [10 ** (-snr / 10) for snr in snrs]
    dev = torch.device("cuda")
    sweep = SecondVolterraV2VAESweep(M, L, l1, l2, 5e-3, const, S, B, noise_variance=nv0, torch_device=dev)
    assert sweep._fit.on_chip
    y_eq, curves = sweep.fit(rxs)
    assert y_eq.shape == (M, nb * B) and curves.shape == (M, nb)
    dec = sweep.decide(rxs)
    ones = []
    for m in range(M):
        one = SecondVolterraV2VAE(channel_memory=L, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                                  noise_variance=nv0[m], torch_device=dev, equaliser_n_lags1=l1, equaliser_n_lags2=l2)
        one.initialize_optimizer()
        ye, lc = one.fit(rxs[m])
        assert np.array_equal(ye, y_eq[m]) and np.array_equal(lc, curves[m])
        assert torch.equal(one._pack, sweep.params[m])
        assert torch.equal(one._noise_var[0], sweep.noise_variance[m])
        q = one.estimate_symbol_probs(rxs[m])
        assert np.array_equal(q.argmax(axis=0), dec[m])
        ones.append(one)
    # A second fit() continues the optimiser state AND the learning-rate decay: the reference builds a new StepLR on the same
    # optimiser, which starts from the already decayed rate (1 % of the initial one after the first call).
    y2, c2 = sweep.fit(rxs)
    assert np.isfinite(c2).all() and sweep.step_count == 2 * nb
    assert abs(sweep._current_lr / 5e-3 - 1e-4) < 1e-9
    for m, one in enumerate(ones):
        ye, lc = one.fit(rxs[m])
        assert abs(one.optimizer.param_groups[0]["lr"] - sweep._current_lr) < 1e-12
        assert np.array_equal(ye, y2[m]) and np.array_equal(lc, c2[m])
        assert torch.equal(one._pack, sweep.params[m])


def test_large_sweep_more_workgroups_than_compute_units_equals_individual_fits():
    """640 models in one launch - more workgroups than the 256 compute units hold at once, so workgroups of later models start
    as earlier ones retire - on 16 distinct sequences shared round-robin (the layout of the benchmark's sweep workload); a
    sample of the models, among them the first, the last and those around the 256 / 512 boundaries, must equal individual
    drop-in fits bit for bit (lib/real_equalization.py:482-502 once per model, main_v2vae.py:75-91)."""
    from volterra_vae_amd.real_equalization import SecondVolterraV2VAE
    from volterra_vae_amd.sweep import SecondVolterraV2VAESweep
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    S, B, L, l1, l2, nb, M = 2, 500, 15, 25, 25, 10, 640
    const = pam_constellation(4)
    snrs = np.linspace(10.0, 25.0, 16)
    base = np.stack([nonlinear_isi(nb * B, const, S, snr, np.random.default_rng(300 + i))[0] for i, snr in enumerate(snrs)])
    idx = np.arange(M) % 16
    nv0 = 10 ** (-snrs[idx] / 10)
    dev = torch.device("cuda")
    sweep = SecondVolterraV2VAESweep(M, L, l1, l2, 5e-3, const, S, B, noise_variance=nv0, torch_device=dev)
    assert sweep._fit.on_chip
    y_eq, curves = sweep.fit(base[idx])
    assert np.isfinite(curves).all() and y_eq.shape == (M, nb * B)
    # models that share a sequence and a starting noise variance are the same computation
    assert np.array_equal(curves[3], curves[3 + 16 * 39]) and np.array_equal(y_eq[3], y_eq[3 + 16 * 39])
    for m in (0, 1, 17, 255, 256, 257, 300, 511, 512, 513, 638, 639):
        one = SecondVolterraV2VAE(channel_memory=L, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                                  noise_variance=float(nv0[m]), torch_device=dev, equaliser_n_lags1=l1, equaliser_n_lags2=l2)
        one.initialize_optimizer()
        ye, lc = one.fit(base[idx[m]])
        assert np.array_equal(ye, y_eq[m]) and np.array_equal(lc, curves[m]), m
        assert torch.equal(one._pack, sweep.params[m]), m
        assert torch.equal(one._noise_var[0], sweep.noise_variance[m]), m
