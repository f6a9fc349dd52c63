#!/usr/bin/env python
"""Per-stage timing of one training step for an arbitrary shape (complex or real), e.g. the L=65 configuration:
   python tools/time_shape.py --L 65 --l1 65 --l2 65 --order 64 --log2-symbols 20"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volterra_vae_amd.ops import Plan  # noqa: E402
from volterra_vae_amd.synth import nonlinear_isi_range, pam_constellation, qam_constellation  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=65)
ap.add_argument("--l1", type=int, default=65)
ap.add_argument("--l2", type=int, default=65)
ap.add_argument("--S", type=int, default=2)
ap.add_argument("--order", type=int, default=64)
ap.add_argument("--real", action="store_true")
ap.add_argument("--log2-symbols", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda")
B = 1 << a.log2_symbols
ns = 1 if a.real else 2
M = a.order if a.real else int(round(np.sqrt(a.order)))
const = pam_constellation(a.order) if a.real else qam_constellation(a.order)
rx, _ = nonlinear_isi_range(7, const, a.S, 18.0, 0, B, B)
y = torch.from_numpy(np.stack((rx.real, rx.imag)) if not a.real else rx[None, :]).float().to(dev).contiguous()
plan = Plan(ns=ns, sps=a.S, l1=a.l1, pad1=a.l1 // 2, l2=a.l2, L=a.L, C_levels=M, n_glob=B * a.S, device=dev)
n = plan.sizes
o = np.cumsum([0] + n)
params = torch.zeros(sum(n), device=dev)
params[a.L // 2] = 1
params[n[0] + n[1] + a.l1 // 2] = 1
params[o[4]:] = 1
params[o[1]:o[2]] = 0.003 * torch.randn(n[1], device=dev)
h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
grads = torch.zeros_like(params)
lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev)
levels = torch.stack([lv] * ns).contiguous()
prior = torch.full_like(levels, 1.0 / M)
nv = torch.tensor([0.03], device=dev)
xhat = torch.empty((ns, B), device=dev)
stages = {
    "prep_tables": lambda: plan.prep_tables(h1, h2, k1, k2 if a.l2 else None),
    "eq_fwd": lambda: plan.eq_fwd(y, xhat),
    "demap_moments": lambda: plan.demap_moments(xhat, levels, prior, sc, nv),
    "elbo_prepass": lambda: plan.elbo_prepass(),
    "elbo_fused": lambda: plan.elbo_fused(y, prepass=False),
    "reduce_scalars": lambda: (plan.reduce_scalars(), plan.finalize_scalars()),
    "sym_bwd": lambda: plan.sym_bwd(xhat, levels, prior, sc, nv),
    "eq_bwd": lambda: plan.eq_bwd(y, plan.gxhat),
    "finish_grads": lambda: plan.finish_grads(7, grads[o[0]:o[1]], grads[o[1]:o[2]], grads[o[2]:o[3]], grads[o[3]:o[4]] if a.l2 else None, grads[o[4]:o[5]]),
}
ms = {k: 0.0 for k in stages}
reps = 3
for it in range(reps + 1):
    for name, fn in stages.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if it:
            ms[name] += e0.elapsed_time(e1) / reps
tot = sum(ms.values())
print({k: round(v, 3) for k, v in ms.items()})
print(f"step {tot:.2f} ms for 2^{a.log2_symbols} symbols = {B / tot / 1e3:.1f} Msym/s; loss {float(plan.scal[2]):.4e}")
