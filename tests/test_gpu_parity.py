"""GPU parity: every stage of the HIP path, called through the C ABI, against the golden vectors
of the reference (real PAM) and against the CPU oracle (complex QAM; parity unpinned by the
reference, see oracle/v2vae_oracle.py).  Tolerances: 1e-5 relative (norm-wise) in fp32."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
STEP32 = sorted(os.path.basename(p)[len("v2vae_real_step_"):-4]
                for p in glob.glob(os.path.join(GOLDEN, "v2vae_real_step_*f32.npz")))
TOL = 1e-5
TOL_EXACT = 3e-6   # against an exact evaluation of the same float32 inputs (observed: <= 2.1e-6, most <= 4e-7)


def rel(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def run_step(plan, y, h1, h2, k1, k2, levels, prior, scaling, nv, want_q=False):
    """All stages through the C ABI; returns a dict of device tensors."""
    from volterra_vae_amd import _lib
    p = plan.p
    xhat = torch.empty((p.ns, p.sym_cnt), dtype=torch.float32, device="cuda")
    q = torch.empty((p.ns, p.C, p.sym_cnt), dtype=torch.float32, device="cuda") if want_q else None
    plan.prep_tables(h1, h2, k1, k2)
    plan.eq_fwd(y, xhat)
    plan.demap_moments(xhat, levels, prior, scaling, nv, q)
    plan.elbo_fused(y)
    plan.reduce_scalars()
    plan.finalize_scalars()
    gx = plan.sym_bwd(xhat, levels, prior, scaling, nv)
    plan.eq_bwd(y, gx)
    g = dict(h1=torch.empty_like(h1), h2=torch.empty_like(h2), k1=torch.empty_like(k1),
             k2=None if k2 is None else torch.empty_like(k2), sc=torch.empty_like(scaling))
    plan.finish_grads(7, g["h1"], g["h2"], g["k1"], g["k2"], g["sc"])
    torch.cuda.synchronize()
    sc = plan.scal.cpu().numpy()
    return dict(xhat=xhat, q=q, E=sc[_lib.SCAL_E], KL=sc[_lib.SCAL_KL], loss=sc[_lib.SCAL_LOSS], g=g, gx=gx)


def _oracle_on_float32_inputs(g, S, lin):
    """The CPU oracle (fp64 arithmetic; pinned to the reference's fp64 goldens to 1e-11 by tests/test_oracle_golden.py) on the
    inputs rounded to float32, i.e. exactly what the GPU is given."""
    from oracle import v2vae_oracle as orc
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32).astype(np.float64))
    y, levels = t(g["rx"]), t(g["constellation"])
    prior = torch.ones_like(levels) / len(levels)
    h1, h2, sc = t(g["h1"]).requires_grad_(), t(g["h2"]).requires_grad_(), t(g["noise_scaling"]).requires_grad_()
    nv = torch.tensor(float(np.float32(g["noise_variance_in"])), dtype=torch.float64)
    if lin:
        w = t(g["eq_w"]).reshape(-1).requires_grad_()
        xhat = orc.fir_equaliser(y, w, S)
    else:
        k1, k2 = t(g["k1"]).requires_grad_(), t(g["k2"]).requires_grad_()
        xhat = orc.volterra_equaliser(y, k1, k2, S)
    q = orc.soft_demap(xhat, levels, nv, sc)
    loss, E, _ = orc.elbo_reference_form(xhat, y, h1, h2, levels, prior, nv, sc, S)
    loss.backward()
    out = dict(xhat=xhat.detach().numpy(), q=q.detach().numpy(), loss=np.array([float(loss)]), E=np.array([float(E)]),
               g_h1=h1.grad.numpy(), g_h2=h2.grad.numpy(), g_noise_scaling=sc.grad.numpy())
    if lin:
        out["g_eq_w"] = w.grad.numpy()
    else:
        out["g_k1"], out["g_k2"] = k1.grad.numpy(), k2.grad.numpy()
    return out


@pytest.mark.parametrize("name", [n for n in STEP32 if "S3" not in n])
def test_real_step_on_matrix_core_elbo_kernel_matches_reference_golden(name):
    """The real model through the matrix-core ELBO kernel (a zero quadrature stream plus the diagonal-quirk terms; the default from
    2^12 symbols on, forced here): the kernel the complex headline shape runs, pinned by the reference's own goldens."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    g32 = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name}.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name[:-3]}f64.npz"))
    S, L, l1, l2, B = (int(g32[k]) for k in ("S", "L", "l1", "l2", "B"))
    lin = bool(int(g32["linear_eq"]))
    levels = dev(g32["constellation"][None, :])
    prior = torch.ones_like(levels) / levels.shape[1]
    y = dev(g32["rx"][None, :])
    outs = {}
    for tag, flags in (("mfma", _lib.FLAG_REAL_MFMA), ("generic", 0)):
        plan = Plan(ns=1, sps=S, l1=l1, pad1=(l1 - 1) // 2 if lin else l1 // 2, l2=0 if lin else l2, L=L,
                    C_levels=levels.shape[1], n_glob=B * S, device="cuda", flags=flags)
        k1 = dev(g32["eq_w"].reshape(-1)) if lin else dev(g32["k1"])
        k2 = None if lin else dev(g32["k2"])
        nv = dev(np.array([float(g32["noise_variance_in"])]))
        outs[tag] = run_step(plan, y, dev(g32["h1"]), dev(g32["h2"]), k1, k2, levels, prior, dev(g32["noise_scaling"][None, :]), nv, True)
    out = outs["mfma"]
    def chk(ours, key, flat=False):
        r64 = np.asarray(g64[key], dtype=np.float64)
        if flat:
            r64 = r64.reshape(-1)
        e64 = rel(ours, r64)
        print(f"{name} {key}: matrix-core path vs fp64 ref {e64:.2e}")
        assert e64 < TOL, f"{key}: {e64:.2e} vs fp64 reference"
    chk(np.array([out["loss"]]), "loss")
    chk(np.array([out["E"]]), "E")
    chk(out["g"]["h1"], "g_h1")
    chk(out["g"]["h2"], "g_h2")
    if not int(g32["no_scaling"]):
        chk(out["g"]["sc"][0], "g_noise_scaling")
    if lin:
        chk(out["g"]["k1"], "g_eq_w", flat=True)
    else:
        chk(out["g"]["k1"], "g_k1")
        chk(out["g"]["k2"], "g_k2")
    # and the two ELBO kernels against each other
    gen = outs["generic"]
    assert abs(out["E"] - gen["E"]) < 2e-6 * abs(gen["E"])
    for k in ("h1", "h2", "k1", "sc"):
        assert rel(out["g"][k], gen["g"][k].cpu().numpy()) < 5e-6, k
    assert rel(out["gx"], gen["gx"].cpu().numpy()) < 5e-6


@pytest.mark.parametrize("L,S,l1,l2,B,M", [(15, 2, 25, 25, 40000, 4), (25, 2, 9, 5, 33000, 4), (13, 1, 5, 3, 50000, 2),
                                           (15, 2, 5, 7, 6000, 4), (11, 2, 13, 17, 5000, 4), (9, 1, 3, 21, 4500, 2),
                                           (25, 2, 31, 31, 7000, 4), (13, 2, 25, 41, 5000, 4), (7, 3, 1, 1, 4200, 4),
                                           (15, 2, 9, 79, 4100, 4), (15, 2, 33, 65, 4200, 4), (9, 1, 47, 47, 4300, 2)])
def test_real_large_batch_default_is_matrix_core_elbo_and_agrees_with_generic(L, S, l1, l2, B, M):
    """From 2^12 owned symbols on the real model runs on the matrix-core kernels by default; the generic kernels (reference-pinned
    by the goldens) on the same inputs are the check.  The shapes reach every table width of the symmetric-form forward
    equaliser (up to 9 / 17 / 25 / 31 / 47 / 65 lags), second-order kernels beyond it (generic forward, matrix-core backward) and a
    first-order kernel longer than the second-order one (generic equaliser, matrix-core ELBO)."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    rng = np.random.default_rng(21)
    rx, _ = nonlinear_isi(B, pam_constellation(M), S, 17.0, rng)
    y = dev(np.real(rx)[None, :])
    lv = np.arange(-(M - 1), M, 2).astype(float)[None, :]
    levels, prior = dev(lv), dev(np.ones_like(lv) / M)
    h1 = 0.1 * rng.standard_normal(L); h1[L // 2] += 1.0
    h2 = 0.02 * rng.standard_normal((L, L))
    k1 = 0.1 * rng.standard_normal(l1); k1[l1 // 2] += 1.0
    k2 = 0.01 * rng.standard_normal((l2, l2))
    sc = 1.0 + 0.2 * rng.random((1, M))
    outs = []
    for flags in (0, _lib.FLAG_GENERIC):
        plan = Plan(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", flags=flags)
        o = run_step(plan, y, dev(h1), dev(h2), dev(k1), dev(k2), levels, prior, dev(sc), dev(np.array([0.06])), want_q=True)
        outs.append(dict(E=float(o["E"]), loss=float(o["loss"]), gx=o["gx"].clone(), xhat=o["xhat"].clone(), q=o["q"].clone(),
                         **{k: v.clone() for k, v in o["g"].items()}))
    a, b = outs
    assert abs(a["E"] - b["E"]) < 2e-6 * abs(b["E"]) and abs(a["loss"] - b["loss"]) < 2e-6 * abs(b["loss"])
    for k in ("gx", "h1", "h2", "k1", "k2", "sc"):
        assert rel(a[k], b[k].cpu().numpy()) < 5e-6, k
    # what inference returns (equalised symbols, posteriors) comes from the same two stages
    assert rel(a["xhat"], b["xhat"].cpu().numpy()) < 2e-6
    assert float((a["q"] - b["q"]).abs().max()) < 2e-5


@pytest.mark.parametrize("name", STEP32)
def test_real_step_matches_reference_golden(name):
    from volterra_vae_amd.ops import Plan
    g32 = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name}.npz"))
    g64 = np.load(os.path.join(GOLDEN, f"v2vae_real_step_{name[:-3]}f64.npz"))
    S, L, l1, l2, B = (int(g32[k]) for k in ("S", "L", "l1", "l2", "B"))
    lin = bool(int(g32["linear_eq"]))
    levels = dev(g32["constellation"][None, :])
    prior = torch.ones_like(levels) / levels.shape[1]
    y = dev(g32["rx"][None, :])
    plan = Plan(ns=1, sps=S, l1=l1, pad1=(l1 - 1) // 2 if lin else l1 // 2, l2=0 if lin else l2, L=L,
                C_levels=levels.shape[1], n_glob=B * S, device="cuda")
    k1 = dev(g32["eq_w"].reshape(-1)) if lin else dev(g32["k1"])
    k2 = None if lin else dev(g32["k2"])
    nv = dev(np.array([float(g32["noise_variance_in"])]))
    out = run_step(plan, y, dev(g32["h1"]), dev(g32["h2"]), k1, k2, levels, prior, dev(g32["noise_scaling"][None, :]), nv, True)
    exact = _oracle_on_float32_inputs(g32, S, lin)
    # Against the fp64 run of the reference: 1e-5 relative.  Against its fp32 run, the reference's own
    # fp32 rounding (its distance to the fp64 run, up to 1e-5 on the equaliser gradients) is added.
    def chk(ours, key, flat=False):
        r64 = np.asarray(g64[key], dtype=np.float64)
        r32 = np.asarray(g32[key], dtype=np.float64)
        if flat:
            r64, r32 = r64.reshape(-1), r32.reshape(-1)
        own = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        e64, e32 = rel(ours, r64), rel(ours, r32)
        print(f"{name} {key}: vs fp64 ref {e64:.2e}, vs fp32 ref {e32:.2e} (ref32-ref64 {own:.2e})")
        assert e64 < TOL, f"{key}: {e64:.2e} vs fp64 reference"
        assert e32 < TOL + own, f"{key}: {e32:.2e} vs fp32 reference (reference fp32-fp64 distance {own:.2e})"
        # Exact (fp64) evaluation of the SAME float32-rounded inputs by the oracle: what is left of the distance to the fp64
        # reference above is the rounding of samples and parameters to float32 on the way in (amplified by
        # 2 |x - a| / noise variance in the soft demapper), not kernel arithmetic.
        ex = rel(ours, exact[key].reshape(-1) if flat else exact[key])
        print(f"{name} {key}: vs fp64 oracle on the float32 inputs {ex:.2e}")
        assert ex < TOL_EXACT, f"{key}: {ex:.2e} vs the fp64 oracle on the same inputs"

    chk(out["xhat"][0], "xhat")
    chk(out["q"][0], "q")
    chk(np.array([out["loss"]]), "loss")
    chk(np.array([out["E"]]), "E")
    chk(out["g"]["h1"], "g_h1")
    chk(out["g"]["h2"], "g_h2")
    if not int(g32["no_scaling"]):
        chk(out["g"]["sc"][0], "g_noise_scaling")
    if lin:
        chk(out["g"]["k1"], "g_eq_w", flat=True)
    else:
        chk(out["g"]["k1"], "g_k1")
        chk(out["g"]["k2"], "g_k2")


def _complex_inputs(seed, L, S, l1, l2, B, M):
    rng = np.random.default_rng(seed)
    from volterra_vae_amd.synth import nonlinear_isi, qam_constellation
    rx, _ = nonlinear_isi(B, qam_constellation(M * M), S, 16.0, rng)
    y = np.stack((rx.real, rx.imag))
    lv = np.arange(-(M - 1), M, 2).astype(float)
    levels = np.stack((lv, lv))
    prior = np.ones_like(levels) / M
    scaling = 1.0 + 0.2 * rng.random((2, M))
    h1 = 0.1 * rng.standard_normal((2, L)); h1[0, L // 2] += 1.0
    h2 = 0.03 * rng.standard_normal((2, L, L))
    k1 = 0.1 * rng.standard_normal((2, l1)); k1[0, l1 // 2] += 1.0
    k2 = 0.02 * rng.standard_normal((2, l2, l2)) if l2 else None
    return y, levels, prior, scaling, h1, h2, k1, k2


@pytest.mark.parametrize("L,S,l1,l2,B,M", [(7, 2, 5, 3, 300, 4), (6, 1, 3, 0, 257, 2), (9, 3, 7, 5, 600, 4),
                                           (25, 2, 25, 25, 1500, 4), (15, 2, 11, 9, 4000, 8), (5, 1, 1, 1, 70, 2),
                                           (4, 2, 1, 3, 513, 2), (31, 2, 31, 31, 1025, 4),
                                           # the largest level count (256-QAM) and sample counts per symbol up to the limit of 8
                                           (13, 2, 7, 5, 500, 16), (9, 4, 5, 3, 300, 4), (17, 8, 9, 5, 200, 2),
                                           # four samples per symbol on the run ELBO kernel (two waves per phase, 96-window
                                           # strips): 9 and 16 active taps per phase (13- and 16-step instances), several strips
                                           (33, 4, 9, 5, 1300, 4), (61, 4, 7, 3, 900, 2), (25, 4, 25, 25, 700, 4)])
def test_complex_step_matches_oracle(L, S, l1, l2, B, M):
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.ops import Plan
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(3, L, S, l1, l2, B, M)
    nv = 0.08
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tsc = t(h1), t(h2), t(k1), t(scaling)
    tk2 = t(k2) if k2 is not None else None
    ty = torch.tensor(y)
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], None if tk2 is None else tk2[0], None if tk2 is None else tk2[1], S)
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels),
                                         torch.tensor(prior), torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), None if k2 is None else dev(k2), dev(levels), dev(prior),
                   dev(scaling), dev(np.array([nv])))
    assert rel(out["xhat"], xr.detach().numpy()) < TOL
    assert abs(out["loss"] - float(loss.detach())) < TOL * abs(float(loss.detach()))
    assert abs(out["E"] - float(E.detach())) < TOL * abs(float(E.detach()))
    assert abs(out["KL"] - float(kl.detach())) < TOL * abs(float(kl.detach())) + 1e-6
    assert rel(out["g"]["h1"], th1.grad) < TOL
    assert rel(out["g"]["h2"], th2.grad) < TOL
    assert rel(out["g"]["k1"], tk1.grad) < TOL
    assert rel(out["g"]["sc"], tsc.grad) < TOL
    if k2 is not None:
        assert rel(out["g"]["k2"], tk2.grad) < TOL


@pytest.mark.parametrize("L,S,l1,l2,B,M", [(41, 2, 9, 5, 700, 4), (25, 1, 7, 5, 600, 4), (33, 1, 5, 3, 500, 2),
                                           (65, 2, 9, 5, 1100, 4), (95, 2, 7, 0, 800, 4), (33, 2, 5, 3, 900, 4),
                                           (17, 1, 3, 3, 300, 2), (64, 2, 9, 5, 700, 4), (65, 2, 9, 5, 97, 4), (33, 1, 3, 3, 41, 2),
                                           (65, 2, 5, 5, 4500, 2)])
def test_complex_long_channel_blocked_kernel_matches_oracle(L, S, l1, l2, B, M):
    """16 < active taps per phase <= 48: the blocked matrix-core ELBO kernel (2 or 3 blocks of 16 taps, both phase counts; 17 and
    33 taps run as 1 and 2 blocks plus one tap on the side), several tiles and a ragged last one, against the fp64 autograd
    oracle."""
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.ops import Plan
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(8, L, S, l1, l2, B, M)
    h2 *= 0.4
    nv = 0.09
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tsc = t(h1), t(h2), t(k1), t(scaling)
    tk2 = t(k2) if k2 is not None else None
    ty = torch.tensor(y)
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], None if tk2 is None else tk2[0], None if tk2 is None else tk2[1], S)
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels), torch.tensor(prior),
                                         torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), None if k2 is None else dev(k2), dev(levels), dev(prior),
                   dev(scaling), dev(np.array([nv])))
    assert abs(out["loss"] - float(loss.detach())) < TOL * abs(float(loss.detach()))
    assert abs(out["E"] - float(E.detach())) < TOL * abs(float(E.detach()))
    assert rel(out["g"]["h1"], th1.grad) < TOL
    assert rel(out["g"]["h2"], th2.grad) < TOL
    assert rel(out["g"]["k1"], tk1.grad) < TOL
    assert rel(out["g"]["sc"], tsc.grad) < TOL


def test_complex_L65_shape_matches_oracle():
    """64-QAM, channel memory 65, 65 lags (the shape of the 8-GPU configuration) at a size the CPU oracle handles."""
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.ops import Plan
    L, S, l1, l2, B, M = 65, 2, 65, 65, 900, 8
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(6, L, S, l1, l2, B, M)
    h2 *= 0.3
    k2 *= 0.3
    nv = 0.1
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tk2, tsc = t(h1), t(h2), t(k1), t(k2), t(scaling)
    ty = torch.tensor(y)
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], tk2[0], tk2[1], S)
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels), torch.tensor(prior),
                                         torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([nv])))
    assert rel(out["xhat"], xr.detach().numpy()) < TOL
    assert abs(out["loss"] - float(loss.detach())) < TOL * abs(float(loss.detach()))
    for ours, ref in ((out["g"]["h1"], th1.grad), (out["g"]["h2"], th2.grad), (out["g"]["k1"], tk1.grad),
                      (out["g"]["k2"], tk2.grad), (out["g"]["sc"], tsc.grad)):
        assert rel(ours, ref) < TOL


@pytest.mark.parametrize("l1,l2,S,B", [(33, 33, 2, 700), (33, 33, 1, 333), (9, 47, 1, 300), (49, 49, 2, 450), (63, 63, 2, 520),
                                       (65, 65, 2, 1100), (65, 65, 1, 300), (81, 81, 1, 270), (31, 95, 2, 600), (95, 95, 1, 257),
                                       (9, 33, 1, 400), (25, 65, 2, 500), (17, 49, 2, 300)])
def test_complex_long_equaliser_blocked_kernel_matches_oracle(l1, l2, S, B):
    """31 < second-order lags <= 95: the equaliser kernels on the bf16 matrix pipe in split arithmetic (forward: 3 ... 6 blocks of
    16 lags, backward likewise; lag counts of the form 16 k + 1, whose last lag the forward carries on the side of k blocks;
    first-order kernel shorter than or as long as the second-order one; one and two samples per symbol; several tiles and a
    ragged last one) against the fp64 autograd oracle."""
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.ops import Plan
    L, M = 9, 4
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(11, L, S, l1, l2, B, M)
    k2 *= 10.0 / l2
    nv = 0.09
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tk2, tsc = t(h1), t(h2), t(k1), t(k2), t(scaling)
    ty = torch.tensor(y)
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], tk2[0], tk2[1], S)
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels), torch.tensor(prior),
                                         torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([nv])))
    assert rel(out["xhat"], xr.detach().numpy()) < TOL
    assert abs(out["loss"] - float(loss.detach())) < TOL * abs(float(loss.detach()))
    # (the equaliser gradients see the error of xhat amplified by 2 |xhat - level| / (noise variance * scaling) ~ 20 at this
    # noise level: the forward's sum over the lag rows runs in double and is rounded to float once)
    for key, ours, ref in (("k1", out["g"]["k1"], tk1.grad), ("k2", out["g"]["k2"], tk2.grad)):
        e = rel(ours, ref)
        print(f"l1={l1} l2={l2} S={S}: gradient of {key} {e:.2e}")
        assert e < TOL, (key, e)
    assert rel(out["g"]["h1"], th1.grad) < TOL


@pytest.mark.parametrize("L,S,B", [(25, 2, 5000), (25, 2, 331), (16, 1, 2049), (31, 2, 1500), (9, 2, 700), (65, 2, 1300),
                                   (33, 1, 700), (33, 2, 900)])
def test_run_and_block_tile_elbo_kernels_agree(L, S, B):
    """The two matrix-core ELBO kernels of the complex model (every wave on its own run of windows - the default - and the
    earlier block-tile kernel, V2VAE_FLAG_ELBO_TILES) are different programs for the same sums; so are, at 17 and 33 active
    taps per phase, the blocked kernel with one tap carried on the side (the default) and with whole blocks of 16 (the flag)."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    l1 = l2 = 25
    M = 4
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(11, L, S, l1, l2, B, M)
    outs = []
    for flags in (0, _lib.FLAG_ELBO_TILES):
        plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", flags=flags)
        o = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([0.08])))
        outs.append(dict(E=float(o["E"]), loss=float(o["loss"]), gx=o["gx"].clone(), **{k: v.clone() for k, v in o["g"].items()}))
    a, b = outs
    assert abs(a["E"] - b["E"]) < 1e-6 * abs(b["E"]) and abs(a["loss"] - b["loss"]) < 1e-6 * abs(b["loss"])
    for k in ("gx", "h1", "h2", "k1", "k2", "sc"):
        assert rel(a[k], b[k].cpu().numpy()) < 2e-6, k


@pytest.mark.parametrize("L,S,l2,B", [(25, 2, 25, 5000), (25, 2, 25, 333), (9, 1, 15, 2049), (25, 2, 47, 1500), (9, 2, 65, 1300),
                                      (13, 1, 79, 900), (9, 2, 31, 257), (9, 2, 17, 700), (9, 1, 17, 333), (9, 1, 33, 500),
                                      (9, 2, 49, 450), (9, 1, 65, 300), (9, 1, 81, 300)])
def test_split_bf16_and_f32_matrix_routes_agree(L, S, l2, B):
    """Complex model: the equaliser forward and backward (and, up to 16 active taps per phase, the H X and H w products of the run
    ELBO kernel) on bf16 matrix instructions in three-piece split arithmetic - the default - against the same sums on the vector
    pipe / on f32 matrix instructions (V2VAE_FLAG_F32_MATRIX): different programs, both exact to float rounding, at 1 ... 6
    blocks of 16 lags, one and two samples per symbol, tiles that end inside a block of 32 symbols; lag counts 16 k + 1, whose
    last lag (and, in the backward, the column of ones behind it) the bf16 kernels carry on the side of k blocks."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    l1 = l2 if l2 <= 25 else 25
    M = 4
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(13, L, S, l1, l2, B, M)
    k2 *= 10.0 / l2
    outs = []
    for flags in (0, _lib.FLAG_F32_MATRIX):
        plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", flags=flags)
        o = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([0.08])))
        outs.append(dict(E=float(o["E"]), loss=float(o["loss"]), gx=o["gx"].clone(), xhat=o["xhat"].clone(),
                         **{k: v.clone() for k, v in o["g"].items()}))
    a, b = outs
    assert abs(a["E"] - b["E"]) < 1e-6 * abs(b["E"]) and abs(a["loss"] - b["loss"]) < 1e-6 * abs(b["loss"])
    assert rel(a["xhat"], b["xhat"].cpu().numpy()) < 3e-7
    # (the two forwards differ by float rounding of xhat, ~1e-7, which the soft demapper amplifies by 2 |xhat - level| /
    # (noise variance * scaling) ~ 20 on its way into dloss/dxhat and the equaliser gradients)
    for k in ("gx", "h1", "h2", "k1", "k2", "sc"):
        assert rel(a[k], b[k].cpu().numpy()) < (TOL if k in ("gx", "k1", "k2", "sc") else 2e-6), k


@pytest.mark.parametrize("ns,M,B", [(1, 4, 4096), (1, 4, 4099), (2, 4, 6000), (2, 8, 1001), (1, 2, 777), (2, 16, 2048), (1, 3, 5000)])
def test_hard_decisions_equal_argmax_of_posteriors(ns, M, B):
    """v2vae_decide against the arg-max of the soft demapper's posteriors (uniform prior, per-level noise scaling) and against
    a direct evaluation of the reference's logits -(x - c)^2 / (sigma^2 s_c) (lib/real_equalization.py:816-823) in numpy;
    lengths that are and are not multiples of four (16-byte and element-wise accesses)."""
    from volterra_vae_amd.ops import Plan
    rng = np.random.default_rng(5)
    lv = np.arange(-(M - 1), M, 2).astype(np.float32)
    levels = dev(np.stack([lv] * ns))
    prior = torch.full_like(levels, 1.0 / M)
    sc = dev(1.0 + 0.3 * rng.random((ns, M)))
    nv = dev(np.array([0.2]))
    x = (rng.choice(lv, size=(ns, B)) + 0.6 * rng.standard_normal((ns, B))).astype(np.float32)
    xhat = dev(x)
    plan = Plan(ns=ns, sps=2, l1=5, pad1=2, l2=3, L=7, C_levels=M, n_glob=B * 2, device="cuda")
    q = torch.empty((ns, M, B), dtype=torch.float32, device="cuda")
    plan.demap_moments(xhat, levels, prior, sc, nv, q)
    dec = plan.decide(xhat, levels, sc, nv)
    torch.cuda.synchronize()
    assert dec.dtype == torch.int32 and dec.shape == (ns, B)
    scn = sc.cpu().numpy().astype(np.float64)
    z = -(x[:, None, :].astype(np.float64) - lv[None, :, None]) ** 2 / (0.2 * scn[:, :, None])
    ref = np.argmax(z, axis=1)
    d = dec.cpu().numpy()
    # float32 logits can order two levels differently from float64 ones only where they tie to rounding
    differ = d != ref
    if differ.any():
        zs = np.sort(z, axis=1)
        gap = (zs[:, -1, :] - zs[:, -2, :])[differ]
        assert np.all(gap < 1e-5 * np.abs(zs[:, -1, :][differ]) + 1e-6)
    assert differ.mean() < 1e-4
    qa = torch.argmax(q, dim=1).cpu().numpy()
    agree = (d == qa).mean()
    assert agree > 1 - 1e-4, agree


@pytest.mark.parametrize("L,S,l1,l2,B,M,mfma", [(9, 2, 7, 5, 400, 16, False), (11, 4, 9, 5, 300, 8, False), (17, 8, 5, 3, 150, 2, False),
                                                (15, 2, 63, 61, 700, 4, False), (7, 1, 3, 129, 600, 4, False),
                                                (6, 3, 255, 0, 500, 4, False),
                                                # the matrix-core route of the real model with the symmetric-form forward (sums
                                                # in double) at its longest kernels: 47- and 65-lag instances
                                                (15, 2, 41, 41, 700, 4, True), (15, 2, 25, 47, 600, 4, True),
                                                (13, 1, 65, 65, 500, 4, True), (15, 2, 9, 63, 650, 2, True),
                                                # four samples per symbol on the matrix-core route (run ELBO kernel, real instance)
                                                (29, 4, 9, 5, 900, 4, True)])
def test_real_step_at_the_limits_of_the_parameter_ranges_matches_oracle(L, S, l1, l2, B, M, mfma):
    """Real model where the reference's goldens do not reach: PAM-16 / PAM-8 (the 16- and 8-level instantiations), 4 and 8
    samples per symbol, equalisers of 63 / 61 lags, the longest second-order (129 lags) and first-order (255 taps) kernels, an
    even channel memory.  Checked against the fp64 oracle (itself pinned to the reference's fp64 goldens to 1e-11) evaluated on
    the float32-rounded inputs."""
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.ops import Plan
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    rng = np.random.default_rng(31)
    rx, _ = nonlinear_isi(B, pam_constellation(M), S, 18.0, rng)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    y = f32(np.real(rx))
    lv = f32(np.arange(-(M - 1), M, 2))
    h1 = 0.05 * rng.standard_normal(L); h1[L // 2] += 1.0
    h2 = 0.01 * rng.standard_normal((L, L))
    k1 = 0.02 * rng.standard_normal(l1); k1[l1 // 2] += 1.0
    k2 = 0.003 * rng.standard_normal((l2, l2)) / max(l2, 1) ** 0.5 if l2 else None
    sc = 1.0 + 0.2 * rng.random(M)
    h1, h2, k1, sc = f32(h1), f32(h2), f32(k1), f32(sc)
    k2 = None if k2 is None else f32(k2)
    nvf = np.float32(0.15)
    t = lambda a: torch.from_numpy(a.astype(np.float64))
    th1, th2, tk1, tsc = (t(a).requires_grad_() for a in (h1, h2, k1, sc))
    tk2 = None if k2 is None else t(k2).requires_grad_()
    ty, tlv = t(y), t(lv)
    prior = torch.ones_like(tlv) / M
    nv = torch.tensor(float(nvf), dtype=torch.float64)
    xr = orc.fir_equaliser(ty, tk1, S) if k2 is None else orc.volterra_equaliser(ty, tk1, tk2, S)
    loss, E, _ = orc.elbo_reference_form(xr, ty, th1, th2, tlv, prior, nv, tsc, S)
    loss.backward()
    lin = k2 is None
    from volterra_vae_amd import _lib
    plan = Plan(ns=1, sps=S, l1=l1, pad1=(l1 - 1) // 2 if lin else l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda",
                flags=_lib.FLAG_REAL_MFMA if mfma else 0)
    out = run_step(plan, dev(y[None, :]), dev(h1), dev(h2), dev(k1), None if lin else dev(k2), dev(lv[None, :]),
                   dev(np.ones((1, M)) / M), dev(sc[None, :]), dev(np.array([nvf])))
    assert rel(out["xhat"][0], xr.detach().numpy()) < TOL_EXACT
    loss_ref, e_ref = float(loss.detach()), float(E.detach())
    assert abs(out["loss"] - loss_ref) < TOL_EXACT * abs(loss_ref)
    assert abs(out["E"] - e_ref) < TOL_EXACT * abs(e_ref)
    for key, ours, ref in (("h1", out["g"]["h1"], th1.grad), ("h2", out["g"]["h2"], th2.grad), ("k1", out["g"]["k1"], tk1.grad),
                           ("noise scaling", out["g"]["sc"][0], tsc.grad)):
        e = rel(ours, ref.numpy())
        print(f"L={L} S={S} l1={l1} l2={l2} M={M} gradient of {key}: {e:.2e}")
        assert e < TOL, (key, e)
    if not lin:
        assert rel(out["g"]["k2"], tk2.grad.numpy()) < TOL
