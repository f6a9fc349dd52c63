#!/usr/bin/env python
"""Diagnostic (GPU box): where the distance between the complex long-equaliser gradients and the fp64 autograd oracle comes
from.  For each shape of tests/test_gpu_parity.py::test_complex_long_equaliser_blocked_kernel_matches_oracle it prints the
relative error of xhat, loss and every gradient against (a) the oracle on the float64 inputs and (b) the oracle on the same
inputs rounded to float32, i.e. what the GPU is given."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_parity import _complex_inputs, dev, rel, run_step  # noqa: E402
from oracle import v2vae_oracle as orc  # noqa: E402
from volterra_vae_amd.ops import Plan  # noqa: E402


def oracle(y, levels, prior, scaling, h1, h2, k1, k2, nv, S, round32):
    r = (lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)) if round32 else (lambda a: np.asarray(a, dtype=np.float64))
    t = lambda a: torch.tensor(r(a), dtype=torch.float64, requires_grad=True)
    th1, th2, tk1, tk2, tsc = t(h1), t(h2), t(k1), t(k2), t(scaling)
    ty = torch.tensor(r(y))
    xr = orc.complex_equaliser(ty, tk1[0], tk1[1], tk2[0], tk2[1], S)
    nvt = torch.tensor(float(np.float32(nv)) if round32 else nv, dtype=torch.float64)
    loss, E, kl = orc.complex_v2vae_loss(xr, ty, th1[0], th1[1], th2[0], th2[1], torch.tensor(r(levels)), torch.tensor(r(prior)),
                                         nvt, tsc, S)
    loss.backward()
    return dict(xhat=xr.detach().numpy(), loss=float(loss), h1=th1.grad.numpy(), h2=th2.grad.numpy(), k1=tk1.grad.numpy(),
                k2=tk2.grad.numpy(), sc=tsc.grad.numpy())


def main():
    shapes = [(33, 33, 2, 700), (9, 47, 1, 300), (63, 63, 2, 520), (65, 65, 2, 1100), (31, 95, 2, 600), (95, 95, 1, 257),
              (25, 25, 2, 1500)]
    for l1, l2, S, B in shapes:
        L, M = 9, 4
        y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(11, L, S, l1, l2, B, M)
        k2 *= 10.0 / l2
        nv = 0.09
        plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
        out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([nv])))
        # the same step with the oracle's xhat (rounded to float32: 3e-8 relative) in place of the forward kernel's
        o32 = oracle(y, levels, prior, scaling, h1, h2, k1, k2, nv, S, True)
        xin = dev(o32["xhat"])
        dy, dl, dp, dsc, dnv = dev(y), dev(levels), dev(prior), dev(scaling), dev(np.array([nv]))
        plan.prep_tables(dev(h1), dev(h2), dev(k1), dev(k2))
        plan.demap_moments(xin, dl, dp, dsc, dnv)
        plan.elbo_fused(dy)
        plan.reduce_scalars()
        plan.finalize_scalars()
        gx = plan.sym_bwd(xin, dl, dp, dsc, dnv)
        plan.eq_bwd(dy, gx)
        g = dict(h1=torch.empty_like(dev(h1)), h2=torch.empty_like(dev(h2)), k1=torch.empty_like(dev(k1)), k2=torch.empty_like(dev(k2)),
                 sc=torch.empty_like(dsc))
        plan.finish_grads(7, g["h1"], g["h2"], g["k1"], g["k2"], g["sc"])
        torch.cuda.synchronize()
        print(f"l1={l1} l2={l2} S={S} B={B} oracle xhat injected:           " +
              "  ".join(f"g_{k} {rel(g[k], o32[k]):.2e}" for k in ("h1", "h2", "k1", "k2", "sc")), flush=True)
        for tag, r32 in (("f64 inputs", False), ("f32-rounded inputs", True)):
            o = oracle(y, levels, prior, scaling, h1, h2, k1, k2, nv, S, r32)
            line = [f"xhat {rel(out['xhat'], o['xhat']):.2e}", f"loss {abs(out['loss'] - o['loss']) / abs(o['loss']):.2e}"]
            line += [f"g_{k} {rel(out['g'][k], o[k]):.2e}" for k in ("h1", "h2", "k1", "k2", "sc")]
            print(f"l1={l1} l2={l2} S={S} B={B} vs oracle on {tag}: " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
