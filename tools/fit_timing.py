#!/usr/bin/env python
"""Wall time of fit() on the reference's default configuration (real PAM-4, S=2, B=500, L=15, l1=l2=25):
persistent kernel (one launch per fit) vs one fused kernel sequence per batch; and a sweep of models."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from volterra_vae_amd.ops import PersistentFit  # noqa: E402
from volterra_vae_amd.real_equalization import SecondVolterraV2VAE  # noqa: E402
from volterra_vae_amd.synth import nonlinear_isi, pam_constellation  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=2000)
    ap.add_argument("--models", type=int, nargs="*", default=[1, 64, 256, 1024])
    ap.add_argument("--L", type=int, default=15)
    args = ap.parse_args()
    S, B, L, l1, l2, C = 2, 500, args.L, 25, 25, 4
    const = pam_constellation(C)
    rx, _ = nonlinear_isi(args.batches * B, const, S, 16.0, np.random.default_rng(1))
    dev = torch.device("cuda")
    for persistent in (True, False):
        m = SecondVolterraV2VAE(channel_memory=L, learning_rate=5e-3, constellation=const, samples_per_symbol=S, batch_size=B,
                                noise_variance=10 ** -1.6, torch_device=dev, equaliser_n_lags1=l1, equaliser_n_lags2=l2)
        m.persistent_fit = persistent
        m.loss_print_interval = 10 ** 9
        m.initialize_optimizer()
        m.fit(rx[:20 * B * S])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, lc = m.fit(rx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"fit persistent={persistent}: {dt * 1e3:.1f} ms for {args.batches} batches = {dt / args.batches * 1e6:.1f} us/step, "
              f"{args.batches * B / dt / 1e6:.2f} Msym/s, final loss {lc[-1]:.3f}", flush=True)
    y = torch.from_numpy(rx.astype(np.float32)).view(1, -1).to(dev)
    lv = torch.tensor(const, dtype=torch.float32, device=dev).view(1, -1)
    prior = torch.full_like(lv, 1.0 / C)
    nb = min(args.batches, 200)
    for M in args.models:
        pf = PersistentFit(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=C, batch_symbols=B, n_models=M, device=dev)
        n = pf.sizes
        prm = torch.zeros((M, pf.pack), dtype=torch.float32, device=dev)
        prm[:, L // 2] = 1.0
        prm[:, n[0] + n[1] + l1 // 2] = 1.0
        prm[:, n[0] + n[1] + n[2] + n[3]:] = 1.0
        st = [torch.zeros_like(prm) for _ in range(4)]
        nv = torch.full((M,), 10 ** -1.6, dtype=torch.float32, device=dev)
        kw = dict(y=y, levels=lv, prior=prior, params=prm, grads=st[0], adam_m=st[1], adam_v=st[2], adam_vmax=st[3], noise_var=nv,
                  lrs=[5e-3] * nb)
        pf.run(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pf.run(step0=nb, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"sweep of {M} models x {nb} batches: {dt * 1e3:.1f} ms = {dt / nb * 1e6:.1f} us per batch step of all models, "
              f"{M * nb * B / dt / 1e6:.1f} Msym/s aggregate", flush=True)


if __name__ == "__main__":
    main()
