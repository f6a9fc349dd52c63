"""The registered torch custom ops (torch.ops.v2vae.*): schema / fake-tensor / autograd registration checks and values
against the stage-by-stage C-ABI calls."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(ns):
    from volterra_vae_amd.ops import Plan
    from test_gpu_parity import _complex_inputs, dev
    if ns == 2:
        L, S, l1, l2, B, M = 25, 2, 25, 25, 700, 4
        y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(5, L, S, l1, l2, B, M)
    else:
        L, S, l1, l2, B, M = 15, 2, 25, 25, 500, 4
        rng = np.random.default_rng(6)
        from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
        rx, _ = nonlinear_isi(B, pam_constellation(M), S, 16.0, rng)
        y = rx[None, :]
        levels = pam_constellation(M)[None, :]
        prior = np.ones_like(levels) / M
        scaling = 1.0 + 0.2 * rng.random((1, M))
        h1 = 0.1 * rng.standard_normal(L); h1[L // 2] += 1
        h2 = 0.03 * rng.standard_normal((L, L))
        k1 = 0.1 * rng.standard_normal(l1); k1[l1 // 2] += 1
        k2 = 0.02 * rng.standard_normal((l2, l2))
    plan = Plan(ns=ns, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    t = {k: dev(v) for k, v in dict(y=y, levels=levels, prior=prior, scaling=scaling, h1=h1, h2=h2, k1=k1, k2=k2).items()}
    t["nv"] = dev(np.array([0.07]))
    return plan, t


@pytest.mark.parametrize("ns", [1, 2], ids=["real", "complex"])
def test_registered_ops_match_the_stage_calls(ns):
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import run_step, rel
    plan, t = _setup(ns)
    ref = run_step(plan, t["y"], t["h1"], t["h2"], t["k1"], t["k2"], t["levels"], t["prior"], t["scaling"], t["nv"])
    ref_x, ref_loss, ref_g = ref["xhat"].clone(), float(ref["loss"]), {k: v.clone() for k, v in ref["g"].items()}
    p = {k: t[k].clone().requires_grad_() for k in ("h1", "h2", "k1", "k2", "scaling")}
    xhat = torch.ops.v2vae.equaliser(t["y"], p["k1"], p["k2"], plan.handle)
    assert torch.equal(xhat, ref_x)
    loss, E, snap = torch.ops.v2vae.elbo_loss(xhat, t["y"], p["h1"], p["h2"], p["scaling"], t["nv"], t["levels"], t["prior"], plan.handle)
    assert loss.dim() == 0 and loss.grad_fn is not None
    assert abs(float(loss) - ref_loss) <= 1e-7 * abs(ref_loss)
    assert snap.numel() == p["h1"].numel() + p["h2"].numel() + p["scaling"].numel() + 1
    loss.backward()
    for k, r in (("h1", "h1"), ("h2", "h2"), ("k1", "k1"), ("k2", "k2"), ("scaling", "sc")):
        assert p[k].grad.shape == p[k].shape
        assert rel(p[k].grad, ref_g[r].cpu().numpy()) < 1e-6, k


def test_opcheck_schema_fake_and_autograd_registration():
    plan, t = _setup(1)
    k1, k2 = t["k1"].clone().requires_grad_(), t["k2"].clone().requires_grad_()
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.v2vae.equaliser.default, (t["y"], k1, k2, plan.handle), test_utils=tests)
    xhat = torch.ops.v2vae.equaliser(t["y"], k1, k2, plan.handle).detach().requires_grad_()
    h1, h2, sc = (t[k].clone().requires_grad_() for k in ("h1", "h2", "scaling"))
    torch.library.opcheck(torch.ops.v2vae.elbo_loss.default,
                          (xhat, t["y"], h1, h2, sc, t["nv"], t["levels"], t["prior"], plan.handle), test_utils=tests)


def test_released_plan_is_refused():
    from volterra_vae_amd.ops import plan_from_handle
    plan, t = _setup(1)
    h = plan.handle
    assert plan_from_handle(h) is plan
    del plan
    import gc
    gc.collect()
    with pytest.raises(RuntimeError, match="not alive"):
        torch.ops.v2vae.equaliser(t["y"], t["k1"], t["k2"], h)
