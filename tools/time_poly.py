"""ELBO kernel time of the memory-polynomial channel model (real, mode 2) at a large batch:  python tools/time_poly.py [log2 symbols]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from volterra_vae_amd.ops import Plan
from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
B = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
L, S, l1, l2, M = 15, 2, 25, 25, 4
rx, _ = nonlinear_isi(B, pam_constellation(M), S, 16.0, np.random.default_rng(0))
y = torch.from_numpy(rx[None, :].astype(np.float32)).cuda()
for mode in (2, 0):
    plan = Plan(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", mode=mode)
    n = plan.sizes
    g = torch.Generator().manual_seed(1)
    params = (0.02 * torch.randn(sum(n), generator=g)).cuda()
    params[L // 2] += 1; params[n[0] + n[1] + l1 // 2] += 1; params[-M:] = 1
    o = np.cumsum([0] + n)
    h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device="cuda")[None, :].contiguous()
    prior = torch.full_like(lv, 1 / M); nv = torch.tensor([0.05], device="cuda")
    xhat = torch.empty((1, B), device="cuda")
    plan.prep_tables(h1, h2, k1, k2); plan.eq_fwd(y, xhat); plan.demap_moments(xhat, lv, prior, sc, nv)
    for _ in range(3): plan.elbo_fused(y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.elbo_fused(y)
    e1.record(); e1.synchronize()
    print(f"mode {mode}: elbo_fused {e0.elapsed_time(e1) / 10:.3f} ms for 2^{int(np.log2(B))} symbols")
