"""GPU tests of the drop-in Python surface (same names / call sequences as the reference's scripts)
against golden vectors produced by the reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def make_model(g, device="cuda"):
    from volterra_vae_amd.real_equalization import LinearV2VAE, SecondVolterraV2VAE
    sched = str(g["lr_schedule"])
    kw = dict(channel_memory=int(g["L"]), learning_rate=float(g["lr"]), constellation=g["constellation"],
              samples_per_symbol=int(g["S"]), batch_size=int(g["B"]), dtype=torch.float32,
              noise_variance=float(g["noise_variance_in"]), adaptive_noise_variance=True,
              torch_device=torch.device(device), lr_schedule=None if sched == "none" else sched)
    if int(g["linear_eq"]):
        return LinearV2VAE(equaliser_n_taps=int(g["l1"]), **kw)
    return SecondVolterraV2VAE(equaliser_n_lags1=int(g["l1"]), equaliser_n_lags2=int(g["l2"]), **kw)


@pytest.mark.parametrize("case", ["cfg1", "cfg3", "lin"])
def test_fit_trajectory_matches_reference(case, capsys):
    from volterra_vae_amd.utility import calc_ser_from_probs
    g32 = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{case}_f32.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"v2vae_real_fit_{case}_f64.npz"))
    model = make_model(g32)
    assert model.IS_PROBABILISTIC
    model.initialize_optimizer()
    y_eq, loss_curve = model.fit(g32["rx"])
    printed = capsys.readouterr().out
    assert "Batch 0, Loss:" in printed and "(LR:" in printed
    assert y_eq.dtype == np.float32 and loss_curve.dtype == np.float64 and len(loss_curve) == int(g32["n_batches"])

    def chk(ours, key, tol, flat=False):
        r64, r32 = np.asarray(g64[key], dtype=np.float64), np.asarray(g32[key], dtype=np.float64)
        if flat:
            r64, r32 = r64.reshape(-1), r32.reshape(-1)
        own = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        e64, e32 = rel(ours, r64), rel(ours, r32)
        print(f"{case} {key}: vs fp64 ref {e64:.2e}, vs fp32 ref {e32:.2e} (ref32-ref64 {own:.2e})")
        assert e64 < tol + own and e32 < tol + own, f"{key}: {e64:.2e} / {e32:.2e} (own {own:.2e})"

    # 20 dependent Adam steps: the per-step 1e-5 bound on gradients compounds; parameters that start at zero
    # (second-order kernels) are compared against the reference's own fp32-vs-fp64 spread plus 1e-4.
    chk(loss_curve, "loss_curve", 1e-5)
    chk(y_eq, "y_eq", 1e-5)
    chk(model.channel_h1, "h1", 1e-5)
    chk(model.noise_scaling, "noise_scaling", 1e-5)
    chk(model.channel_h2, "h2", 1e-4)
    if case == "lin":
        chk(model.equaliser.weight, "eq_w", 1e-5, flat=True)
    else:
        chk(model.equaliser.kernel1, "k1", 1e-5)
        chk(model.equaliser.kernel2, "k2", 1e-4)
    assert abs(float(model.noise_variance) - g64["noise_variance_traj"][-1]) < 1e-5 * g64["noise_variance_traj"][-1]

    # inference path: estimate_symbol_probs / apply on the validation sequence, SER as the reference computes it
    q = model.estimate_symbol_probs(g32["rx_val"])
    assert q.shape == g32["q_val"].shape
    assert rel(q, g32["q_val"]) < 1e-4
    y_apply = model.apply(g32["rx_val"])
    assert rel(y_apply, g32["y_apply"]) < 1e-5
    ser, delay = calc_ser_from_probs(q, g32["syms_val"], g32["constellation"])
    assert delay == int(g32["ser_delay"])
    assert abs(ser - float(g32["ser"])) <= 2.0 / len(g32["syms_val"])


def test_refit_sequence_over_a_time_varying_channel_matches_reference(capsys):
    """The change-point loop of the reference (main_time_varying_wh.py:121-149): fit() per chunk with carried optimiser
    state, parameters and noise variance, lr_schedule=None, estimate_symbol_probs + SER after every chunk, the channel's
    first FIR swapped between chunks.  Goldens from the reference (tests/golden/make_golden_refit.py)."""
    from volterra_vae_amd.utility import calc_ser_from_probs
    g32 = np.load(os.path.join(GOLDEN, "v2vae_real_refit_cfg3_f32.npz"))
    g64 = np.load(os.path.join(GOLDEN, "v2vae_real_refit_cfg3_f64.npz"))
    model = make_model(g32)
    model.initialize_optimizer()
    nb = int(g32["n_batches"])
    for c in range(int(g32["n_chunks"])):
        y_eq, loss = model.fit(g32[f"rx_{c}"])
        assert len(loss) == nb

        def chk(ours, key, tol):
            r64, r32 = np.asarray(g64[key], dtype=np.float64), np.asarray(g32[key], dtype=np.float64)
            own = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
            e64, e32 = rel(ours, r64), rel(ours, r32)
            print(f"chunk {c} {key}: vs fp64 ref {e64:.2e}, vs fp32 ref {e32:.2e} (ref32-ref64 {own:.2e})")
            assert e64 < tol + own and e32 < tol + own, f"{key}: {e64:.2e} / {e32:.2e} (own {own:.2e})"

        chk(loss, f"loss_{c}", 1e-5)
        chk(y_eq, f"y_eq_{c}", 1e-5)
        chk(model.channel_h1, f"h1_{c}", 1e-5)
        chk(model.equaliser.kernel1, f"k1_{c}", 1e-5)
        chk(model.noise_scaling, f"noise_scaling_{c}", 1e-5)
        chk(model.channel_h2, f"h2_{c}", 1e-4)
        chk(model.equaliser.kernel2, f"k2_{c}", 1e-4)
        nv = float(g64[f"noise_variance_{c}"])
        assert abs(float(model.noise_variance) - nv) < 1e-5 * nv
        q = model.estimate_symbol_probs(g32[f"rx_val_{c}"])
        assert rel(q, g32[f"q_val_{c}"]) < 1e-4
        ser, delay = calc_ser_from_probs(q, g32[f"syms_val_{c}"], g32["constellation"])
        assert delay == int(g32[f"ser_delay_{c}"])
        assert abs(ser - float(g32[f"ser_{c}"])) <= 2.0 / len(g32[f"syms_val_{c}"])
    # the optimiser's step count ran on across the fits (bias corrections of the third chunk use steps 41..60)
    assert model.optimizer.step_count == int(g32["adam_step"]) == nb * int(g32["n_chunks"])
    printed = capsys.readouterr().out
    assert printed.count("Batch 0, Loss:") == int(g32["n_chunks"])


def test_manual_step_api_equals_fused_fit():
    """forward -> _calculate_loss -> _update_model (autograd route) == the fused fit() step."""
    g = np.load(os.path.join(GOLDEN, "v2vae_real_fit_cfg1_f32.npz"))
    S, B = int(g["S"]), int(g["B"])
    a, b = make_model(g), make_model(g)
    a.initialize_optimizer()
    b.initialize_optimizer()
    nb = 12  # >= 10: with fewer batches StepLR gets step_size 0 and raises, in the reference as here
    rx = g["rx"][:nb * B * S]
    y_eq, losses = a.fit(rx)
    y = torch.from_numpy(rx).float().cuda()
    sched = b._make_lr_schedule(nb)
    for n in range(nb):
        ysl = y[n * B * S:(n + 1) * B * S]
        xhat = b.forward(ysl)
        loss = b._calculate_loss(xhat, ysl)
        assert loss.dim() == 0 and loss.grad_fn is not None
        assert abs(float(loss) - losses[n]) < 1e-6 * abs(losses[n])
        assert rel(xhat, y_eq[n * B:(n + 1) * B]) < 1e-6
        b._update_model(loss)
        sched.step()
    assert rel(b._pack, a._pack.cpu().numpy()) < 1e-6
    assert abs(float(a.noise_variance) - float(b.noise_variance)) < 1e-6 * float(a.noise_variance)


def _grads_of(model):
    ps = [model.channel_h1, model.channel_h2, model.noise_scaling]
    if hasattr(model, "equaliser"):
        ps += [model.equaliser.kernel1, model.equaliser.kernel2]
    else:
        ps += [model.equaliser_k1, model.equaliser_k2]
    out = [p.grad.detach().clone() for p in ps]
    for p in ps:
        p.grad = None
    return out


@pytest.mark.parametrize("cplx", [False, True], ids=["real", "complex"])
def test_backward_survives_other_calls_on_the_same_plan(cplx):
    """Autograd lets a caller run anything between a loss and its backward.  Inference on the same length and a second
    loss before the first backward both reuse the model's plan (workspace); the gradients must not change."""
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation, qam_constellation
    rng = np.random.default_rng(9)
    S, B = 2, 600
    if cplx:
        from volterra_vae_amd.complex_equalization import SecondVolterraV2VAE
        const = qam_constellation(16)
        model = SecondVolterraV2VAE(channel_memory=25, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                                    noise_variance=0.05, torch_device=torch.device("cuda"), equaliser_n_lags1=25,
                                    equaliser_n_lags2=25)
        rx, _ = nonlinear_isi(2 * B, const, S, 16.0, rng)
        y = torch.from_numpy(model.to_real_stacked(rx)).float().cuda()
    else:
        from volterra_vae_amd.real_equalization import SecondVolterraV2VAE
        const = pam_constellation(4)
        model = SecondVolterraV2VAE(channel_memory=15, equaliser_n_lags1=25, equaliser_n_lags2=25, learning_rate=5e-3,
                                    constellation=const, batch_size=B, samples_per_symbol=S, noise_variance=0.05,
                                    torch_device=torch.device("cuda"))
        rx, _ = nonlinear_isi(2 * B, const, S, 16.0, rng)
        y = torch.from_numpy(rx).float().cuda()
    model.adaptive_noise_variance = False   # keep sigma^2 fixed so that the three runs see identical inputs
    with torch.no_grad():
        model.channel_h2 += 0.02 * torch.randn_like(model.channel_h2)
    ya, yb = y[..., :B * S].contiguous(), y[..., B * S:].contiguous()

    def loss_of(ysl):
        return model._calculate_loss(model.forward(ysl), ysl)

    loss_of(ya).backward()
    ga = _grads_of(model)
    loss_of(yb).backward()
    gb = _grads_of(model)
    # (1) inference of the same length between loss and backward
    la = loss_of(ya)
    with torch.no_grad():
        model.forward_probs(yb)
    la.backward()
    for g, r in zip(_grads_of(model), ga):
        assert torch.equal(g, r)
    # (2) two losses, then both backwards (the first one's forward state has been overwritten by the second)
    la, lb = loss_of(ya), loss_of(yb)
    la.backward()
    g1 = _grads_of(model)
    lb.backward()
    g2 = _grads_of(model)
    for g, r in zip(g1 + g2, ga + gb):
        assert torch.equal(g, r)
    # (3) gradient accumulation: one backward of the sum
    (loss_of(ya) + loss_of(yb)).backward()
    for g, ra, rb in zip(_grads_of(model), ga, gb):
        assert rel(g, (ra + rb).cpu().numpy()) < 1e-6


def test_long_fit_ser_matches_reference():
    from volterra_vae_amd.synth import nonlinear_isi
    from volterra_vae_amd.utility import decode_from_probs, find_delay
    g = np.load(os.path.join(GOLDEN, "v2vae_real_fit_long_cfg1_f32.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    rx, _ = nonlinear_isi(int(g["B"]) * int(g["n_batches"]), g["constellation"], int(g["S"]), 16.0, rng)
    rx_val, syms_val = nonlinear_isi(int(g["n_val"]), g["constellation"], int(g["S"]), 16.0, rng)
    model = make_model(g)
    model.initialize_optimizer()
    _, loss_curve = model.fit(rx)
    # 400 dependent Adam steps at lr 5e-3 are a chaotic map: the reference's own algorithm re-run on the CPU in
    # fp32 with another summation order, or in fp64, departs from this golden curve by up to 6-8 % at single
    # steps around batch 70-150 (measured with oracle/v2vae_oracle.py) and re-converges.  Hence the loose
    # curve bound; the tight per-step parity is in test_fit_trajectory_matches_reference.
    e_curve = rel(loss_curve, g["loss_curve"])
    e_tail = rel(loss_curve[-50:], g["loss_curve"][-50:])
    print(f"loss curve rel {e_curve:.2e}, last 50 batches {e_tail:.2e}")
    assert e_curve < 5e-2 and e_tail < 2e-2
    q = model.estimate_symbol_probs(rx_val)
    dec = np.argmax(q, axis=0)
    agree = np.mean(dec == g["decisions"])
    ahat = decode_from_probs(q, g["constellation"])
    d = find_delay(ahat, syms_val)
    ser = np.mean(ahat[10:-10] != np.roll(syms_val, d)[10:-10])
    print(f"decision agreement {agree:.5f}, SER {ser:.4e} vs reference {float(g['ser']):.4e}")
    assert agree > 0.98
    assert abs(ser - float(g["ser"])) < 1e-2


def test_error_conventions():
    from volterra_vae_amd.real_equalization import SecondVolterraV2VAE
    from volterra_vae_amd.torch_components import SecondOrderVolterraSeries
    kw = dict(channel_memory=15, equaliser_n_lags1=25, equaliser_n_lags2=25, learning_rate=5e-3,
              constellation=np.array([-3., -1., 1., 3.]), batch_size=500, samples_per_symbol=2, noise_variance=0.1)
    m = SecondVolterraV2VAE(torch_device=torch.device("cuda"), **kw)
    assert repr(m) == "SecondVolterraV2VAE"
    with pytest.raises(Exception, match="Optimizer has not been initialized yet"):
        m.fit(np.zeros(2000))
    with pytest.raises(RuntimeError, match="GPU"):
        SecondVolterraV2VAE(torch_device=torch.device("cpu"), **kw)
    with pytest.raises(NotImplementedError):
        SecondVolterraV2VAE(torch_device=torch.device("cuda"), dtype=torch.float64, **kw)
    with pytest.raises(AssertionError):
        SecondOrderVolterraSeries(24, 25, 2, torch_device=torch.device("cuda"))
    names = [g["name"] for g in m.get_parameters()]
    assert names == ["channel_h1", "channel_h2", "equaliser", "noise_scale"]
    assert isinstance(m.equaliser.kernel1, torch.nn.Parameter) and m.equaliser.kernel1.shape == (25,)
    assert float(m.equaliser.kernel1[12]) == 1.0 and float(m.channel_h1[7]) == 1.0


def test_standalone_modules_match_oracle():
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.torch_components import FIRfilter, SecondOrderVolterraSeries
    rng = np.random.default_rng(4)
    x = rng.standard_normal(1001)
    xt = torch.from_numpy(x)
    eq = SecondOrderVolterraSeries(9, 7, 2, torch_device=torch.device("cuda"))
    with torch.no_grad():
        eq.kernel1 += torch.from_numpy(0.1 * rng.standard_normal(9)).float().cuda()
        eq.kernel2 += torch.from_numpy(0.05 * rng.standard_normal((7, 7))).float().cuda()
    k1 = eq.kernel1.detach().double().cpu().requires_grad_()
    k2 = eq.kernel2.detach().double().cpu().requires_grad_()
    ref = orc.volterra_equaliser(xt, k1, k2, 2)
    out = eq(xt.float().cuda())
    assert out.shape == ref.shape and rel(out, ref.detach().numpy()) < 1e-5
    wgt = torch.from_numpy(rng.standard_normal(ref.shape[0]))
    (ref * wgt).sum().backward()
    (out * wgt.float().cuda()).sum().backward()
    assert rel(eq.kernel1.grad, k1.grad) < 1e-5 and rel(eq.kernel2.grad, k2.grad) < 1e-5
    for n_taps, stride in ((7, 1), (8, 2), (5, 3)):
        f = FIRfilter(n_taps, stride=stride, torch_device=torch.device("cuda"))
        w = rng.standard_normal(n_taps)
        with torch.no_grad():
            f.weights.copy_(torch.from_numpy(w).float())
        pad = ((n_taps - 1) // 2, (n_taps - n_taps % 2) // 2)
        xp = torch.nn.functional.pad(xt, pad)[None, None, :]
        ref = torch.nn.functional.conv1d(xp, torch.flip(torch.from_numpy(w), (0,))[None, None, :], stride=stride).reshape(-1)
        out = f(xt.float().cuda())
        assert out.shape == ref.shape and rel(out, ref.numpy()) < 1e-5


def test_complex_classes_fit_and_manual_step_agree():
    from volterra_vae_amd.complex_equalization import LinearV2VAE, SecondVolterraV2VAE
    from volterra_vae_amd.synth import nonlinear_isi, qam_constellation
    rng = np.random.default_rng(8)
    const = qam_constellation(16)
    S, B = 2, 400
    rx, _ = nonlinear_isi(3 * B, const, S, 16.0, rng)
    for cls, kw in ((SecondVolterraV2VAE, dict(equaliser_n_lags1=11, equaliser_n_lags2=7)), (LinearV2VAE, dict(equaliser_n_taps=11))):
        mk = lambda: cls(channel_memory=9, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                         noise_variance=0.05, torch_device=torch.device("cuda"), **kw)
        a, b = mk(), mk()
        out = a.fit(rx)
        assert np.iscomplexobj(out) and out.shape == (3 * B,)
        y = torch.from_numpy(b.to_real_stacked(rx)).float().cuda()
        for n in range(3):
            ysl = y[:, n * B * S:(n + 1) * B * S]
            xhat = b.forward(ysl)
            loss = b._calculate_loss(xhat, ysl)
            assert abs(float(loss) - float(a.loss_curve[n])) < 1e-6 * abs(float(loss))
            b._update_model(loss)
        assert rel(b._pack, a._pack.cpu().numpy()) < 1e-6
        q = a.estimate_symbol_probs(rx)
        assert q.shape == (8, 3 * B) and np.allclose(q[:4].sum(0), 1, atol=1e-5) and np.allclose(q[4:].sum(0), 1, atol=1e-5)


def test_failure_detection_counts_steps_with_a_non_finite_or_non_positive_exponent_term():
    """The scalar block keeps a running count of steps whose E was not a positive finite number (V2VAE_SCAL_BAD_STEPS): zero on
    healthy steps, incremented when the samples carry a NaN - through the per-batch sequence and through the on-chip fit
    kernel.  The reference has no check on this path (lib/real_equalization.py:919: the NaN loss just propagates)."""
    from volterra_vae_amd.ops import PersistentFit, Plan
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    from volterra_vae_amd import _lib
    dev = torch.device("cuda")
    S, B, L, l1, l2, M = 2, 500, 15, 25, 25, 4
    rx, _ = nonlinear_isi(3 * B, pam_constellation(M), S, 16.0, np.random.default_rng(4))
    y = torch.from_numpy(rx).float().to(dev).view(1, -1).contiguous()
    plan = Plan(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device=dev)
    n = plan.sizes
    def fresh():
        p = torch.zeros(sum(n), device=dev)
        p[L // 2] = 1.0; p[n[0] + n[1] + l1 // 2] = 1.0; p[-M:] = 1.0
        return p
    params = fresh()
    grads, m, v, vm = (torch.zeros_like(params) for _ in range(4))
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev).view(1, -1)
    prior = torch.full_like(lv, 1.0 / M)
    nv = torch.tensor([0.05], device=dev)
    xhat = torch.empty((1, B), device=dev)
    loss = torch.zeros(1, device=dev)
    kw = dict(levels=lv, prior=prior, params=params, grads=grads, adam_m=m, adam_v=v, adam_vmax=vm, noise_var=nv, xhat=xhat,
              loss_out=loss, lr=5e-3)
    plan.train_step(y=y[:, :B * S].contiguous(), step=1, **kw)
    assert plan.bad_steps() == 0 and bool(torch.isfinite(loss).all())
    ybad = y[:, B * S:2 * B * S].clone()
    ybad[0, 77] = float("nan")
    plan.train_step(y=ybad, step=2, **kw)
    assert plan.bad_steps() == 1
    # on-chip fit kernel: three batches, the second one poisoned
    pf = PersistentFit(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, batch_symbols=B, n_models=1, device=dev)
    assert pf.on_chip
    p2 = fresh().view(1, -1)
    g2, m2, v2, vm2 = (torch.zeros_like(p2) for _ in range(4))
    nv2 = torch.tensor([0.05], device=dev)
    y3 = y.clone()
    pf.run(y=y3, levels=lv, prior=prior, params=p2, grads=g2, adam_m=m2, adam_v=v2, adam_vmax=vm2, noise_var=nv2, lrs=[5e-3] * 3)
    torch.cuda.synchronize()
    assert int(pf.scal[0, _lib.SCAL_BAD_STEPS].item()) == 0
    y3[0, B * S + 5] = float("nan")
    p2.copy_(fresh().view(1, -1))
    for t in (g2, m2, v2, vm2):
        t.zero_()
    nv2.fill_(0.05)
    pf.run(y=y3, levels=lv, prior=prior, params=p2, grads=g2, adam_m=m2, adam_v=v2, adam_vmax=vm2, noise_var=nv2, lrs=[5e-3] * 3)
    torch.cuda.synchronize()
    assert int(pf.scal[0, _lib.SCAL_BAD_STEPS].item()) >= 1
