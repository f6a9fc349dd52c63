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
def test_persistent_fit_is_bitwise_the_per_batch_fit(case):
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
