"""v2vae_train_step_static (no allocation: workspace in the library's static device array) against v2vae_train_step
(caller-owned workspace), through the C ABI: bit-identical outputs on two calls in a row, the second on a smaller
problem (the array still holds the first call's contents), then a refused oversize problem."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(ns, L, l1, l2, M, B, seed, S=2):
    from volterra_vae_amd.ops import Plan
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation, qam_constellation
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(seed)
    if ns == 1:
        rx, _ = nonlinear_isi(B, pam_constellation(M), S, 16.0, rng)
        y = torch.from_numpy(rx[None, :].astype(np.float32)).to(dev).contiguous()
    else:
        rx, _ = nonlinear_isi(B, qam_constellation(M * M), S, 16.0, rng)
        y = torch.from_numpy(np.stack((rx.real, rx.imag)).astype(np.float32)).to(dev).contiguous()
    plan = Plan(ns=ns, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device=dev)
    n = plan.sizes
    g = torch.Generator().manual_seed(seed)
    params = (0.02 * torch.randn(sum(n), generator=g)).to(dev)
    params[L // 2] += 1.0
    params[n[0] + n[1] + l1 // 2] += 1.0
    params[-ns * M:] = 1.0
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev)
    levels = torch.stack([lv] * ns).contiguous()
    prior = torch.full_like(levels, 1.0 / M)
    return plan, y, params, levels, prior


def _run(plan, y, params0, levels, prior, static, steps=2):
    dev = y.device
    params = params0.clone()
    grads, m, v, vm = (torch.zeros_like(params) for _ in range(4))
    nv = torch.tensor([0.05], dtype=torch.float32, device=dev)
    xhat = torch.empty((plan.p.ns, plan.p.sym_cnt), dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    losses = []
    for step in range(1, steps + 1):
        plan.train_step(y=y, levels=levels, prior=prior, params=params, grads=grads, adam_m=m, adam_v=v, adam_vmax=vm,
                        noise_var=nv, xhat=xhat, loss_out=loss, lr=5e-3, step=step, static_scratch=static)
        losses.append(loss.clone())
    torch.cuda.synchronize()
    return dict(params=params, grads=grads, m=m, v=v, vm=vm, nv=nv, xhat=xhat, loss=torch.cat(losses))


CASES = [
    dict(ns=2, L=25, l1=25, l2=25, M=4, B=20000, seed=1),   # complex, matrix-core kernels (first call, larger)
    dict(ns=1, L=15, l1=25, l2=25, M=4, B=500, seed=2),     # real, the reference's default batch (second call, smaller)
]


def test_static_scratch_step_equals_workspace_step_twice_in_a_row():
    from volterra_vae_amd import _lib
    for case in CASES:
        plan, y, params, levels, prior = _case(**case)
        need = _lib.lib.v2vae_static_scratch_bytes(C.byref(plan.p))
        assert 0 < need <= (32 << 20)
        got = _run(plan, y, params, levels, prior, static=True)
        ref = _run(plan, y, params, levels, prior, static=False)
        for k in ref:
            assert torch.isfinite(ref[k]).all(), k
            assert torch.equal(got[k], ref[k]), (case, k, float((got[k] - ref[k]).abs().max()))


def test_static_scratch_rejects_a_problem_beyond_the_array():
    from volterra_vae_amd import _lib
    plan, y, params, levels, prior = _case(ns=2, L=25, l1=25, l2=25, M=4, B=1 << 20, seed=3)
    assert _lib.lib.v2vae_static_scratch_bytes(C.byref(plan.p)) > (32 << 20)
    with pytest.raises(_lib.V2VAEError, match="static array holds"):
        _run(plan, y, params, levels, prior, static=True, steps=1)
    # and a workspace pointer passed by mistake is refused, not ignored
    a = _lib.TrainArgs()
    a.ws = plan.ws.data_ptr()
    assert _lib.lib.v2vae_train_step_static(C.byref(plan.p), C.byref(a), None) == _lib.EINVAL


def test_static_scratch_step_allocates_no_device_memory():
    """What the entry point exists for: after one warm-up call (address of the array resolved, code object loaded) further
    calls leave the free device memory and torch's reserved pool exactly where they were."""
    plan, y, params, levels, prior = _case(**CASES[0])
    dev = y.device
    grads, m, v, vm = (torch.zeros_like(params) for _ in range(4))
    nv = torch.tensor([0.05], dtype=torch.float32, device=dev)
    xhat = torch.empty((plan.p.ns, plan.p.sym_cnt), dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    kw = dict(y=y, levels=levels, prior=prior, params=params, grads=grads, adam_m=m, adam_v=v, adam_vmax=vm, noise_var=nv,
              xhat=xhat, loss_out=loss, lr=5e-3, static_scratch=True)
    plan.train_step(step=1, **kw)
    torch.cuda.synchronize()
    free0, reserved0, allocated0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved(), torch.cuda.memory_allocated()
    for step in range(2, 8):
        plan.train_step(step=step, **kw)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.cuda.memory_reserved() == reserved0 and torch.cuda.memory_allocated() == allocated0
    assert torch.isfinite(loss).all()
