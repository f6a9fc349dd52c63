#!/usr/bin/env python
"""Inference path (equaliser forward, soft demapper with posteriors out, hard decisions) timed stage by stage on one GPU:
   python tools/time_inference.py [--log2-symbols 22]
Shapes: the real PAM-4 model of the reference's default script (L=15, l1=l2=25, sps=2) and the complex 16-QAM headline shape."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volterra_vae_amd.ops import Plan  # noqa: E402
from volterra_vae_amd.synth import nonlinear_isi_range, pam_constellation, qam_constellation  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2-symbols", type=int, default=22)
a = ap.parse_args()
dev = torch.device("cuda")
B = 1 << a.log2_symbols
S = 2
for name, ns, L, l1, l2, order in (("real PAM-4, L=15, l1=l2=25", 1, 15, 25, 25, 4), ("complex 16-QAM, L=25, l1=l2=25", 2, 25, 25, 25, 16)):
    M = order if ns == 1 else int(round(np.sqrt(order)))
    const = pam_constellation(order) if ns == 1 else qam_constellation(order)
    rx, _ = nonlinear_isi_range(7, const, S, 18.0, 0, B, B)
    y = torch.from_numpy(np.stack((rx.real, rx.imag)) if ns == 2 else np.real(rx)[None, :]).float().to(dev).contiguous()
    plan = Plan(ns=ns, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device=dev)
    n = plan.sizes
    o = np.cumsum([0] + n)
    params = torch.zeros(sum(n), device=dev)
    params[L // 2] = 1
    params[n[0] + n[1] + l1 // 2] = 1
    params[o[4]:] = 1
    params[o[3]:o[4]] = 0.003 * torch.randn(n[3], device=dev)
    h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev)
    levels = torch.stack([lv] * ns).contiguous()
    prior = torch.full_like(levels, 1.0 / M)
    nv = torch.tensor([0.03], device=dev)
    xhat = torch.empty((ns, B), device=dev)
    q = torch.empty((ns, M, B), device=dev)
    stages = {
        "prep_tables": lambda: plan.prep_tables(h1, h2, k1, k2),
        "eq_fwd": lambda: plan.eq_fwd(y, xhat),
        "demap (posteriors out)": lambda: plan.demap_moments(xhat, levels, prior, sc, nv, q),
        "decide": lambda: plan.decide(xhat, levels, sc, nv),
    }
    ms = {k: 0.0 for k in stages}
    reps = 5
    for it in range(reps + 1):
        for st, fn in stages.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            if it:
                ms[st] += e0.elapsed_time(e1) / reps
    probs = ms["prep_tables"] + ms["eq_fwd"] + ms["demap (posteriors out)"]
    hard = ms["prep_tables"] + ms["eq_fwd"] + ms["decide"]
    print(f"{name}, 2^{a.log2_symbols} symbols: " + ", ".join(f"{k} {v:.3f} ms" for k, v in ms.items()))
    print(f"   estimate_symbol_probs (equaliser + posteriors): {probs:.3f} ms = {B / probs / 1e6:.2f} Gsym/s;"
          f" apply + hard decisions: {hard:.3f} ms = {B / hard / 1e6:.2f} Gsym/s")
    del plan
