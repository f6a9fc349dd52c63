"""Diagnostic (not collected): error of xhat, of the gradient with respect to xhat and of the equaliser gradients against the
fp64 oracle for the split-bf16 routes, the f32 routes, and the bf16 backward alone."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_parity import _complex_inputs, run_step, dev, rel
from oracle import v2vae_oracle as orc
from volterra_vae_amd import _lib
from volterra_vae_amd.ops import Plan

L, M = 9, 4
for (l1, l2, S, B) in [(33, 33, 2, 1500), (63, 63, 2, 1300), (65, 65, 2, 1300), (95, 95, 1, 1100)]:
    y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(11, L, S, l1, l2, B, M)
    k2 *= 10.0 / l2
    nv = 0.09
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tk2, tsc = t(h1), t(h2), t(k1), t(k2), t(scaling)
    ty = torch.tensor(y)
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], tk2[0], tk2[1], S)
    xr.retain_grad()
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(levels), torch.tensor(prior),
                                         torch.tensor(nv, dtype=torch.float64), tsc, S)
    loss.backward()
    for name, flags in (("bf16", 0), ("f32", _lib.FLAG_F32_MATRIX), ("f32fwd+bf16bwd", _lib.FLAG_EQ_MFMA_FWD)):
        plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", flags=flags)
        out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([nv])))
        xh = out["xhat"].cpu().numpy().astype(np.float64)
        xo = xr.detach().numpy()
        d = xh - xo
        print(f"l2={l2} S={S} {name:16s} xhat max {np.abs(d).max()/np.abs(xo).max():.2e} rms {np.sqrt((d**2).mean())/np.sqrt((xo**2).mean()):.2e} "
              f"mean {d.mean()/np.sqrt((xo**2).mean()):+.2e}  gx {rel(out['gx'], xr.grad):.2e}  k1 {rel(out['g']['k1'], tk1.grad):.2e} k2 {rel(out['g']['k2'], tk2.grad):.2e} "
              f"h1 {rel(out['g']['h1'], th1.grad):.2e}")
