#!/usr/bin/env python
"""Stage-by-stage wall time of one training step inside the on-chip fit kernel.
Build first:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC -DV2VAE_STAMPS -o tools/libstamps_RS.so volterra_vae_amd/csrc/*.hip"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import volterra_vae_amd._lib as L  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.getcwd(), "tools/libstamps_RS.so"))
for name, (res, args) in L.EXPORTS.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
L.lib = lib
import volterra_vae_amd.ops as ops  # noqa: E402

ops.lib = lib
from volterra_vae_amd.synth import nonlinear_isi, pam_constellation  # noqa: E402

S, B, Lc, l1, l2, C, nb = 2, 500, 15, 25, 25, 4, 100
const = pam_constellation(C)
rx, _ = nonlinear_isi(nb * B, const, S, 16.0, np.random.default_rng(1))
dev = torch.device("cuda")
y = torch.from_numpy(rx.astype(np.float32)).view(1, -1).to(dev)
lv = torch.tensor(const, dtype=torch.float32, device=dev).view(1, -1)
prior = torch.full_like(lv, 1.0 / C)
pf = ops.PersistentFit(ns=1, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=Lc, C_levels=C, batch_symbols=B, n_models=1, device=dev)
n = pf.sizes
prm = torch.zeros((1, pf.pack), dtype=torch.float32, device=dev)
prm[:, Lc // 2] = 1.0
prm[:, n[0] + n[1] + l1 // 2] = 1.0
prm[:, n[0] + n[1] + n[2] + n[3]:] = 1.0
st = [torch.zeros_like(prm) for _ in range(4)]
nv = torch.full((1,), 10 ** -1.6, dtype=torch.float32, device=dev)
for _ in range(2):
    pf.run(y=y, levels=lv, prior=prior, params=prm, grads=st[0], adam_m=st[1], adam_v=st[2], adam_vmax=st[3], noise_var=nv, lrs=[5e-3] * nb)
torch.cuda.synchronize()
raw = pf.ws[0, :44].cpu().numpy().view(np.uint64).astype(np.int64)
t = raw[:12]
print(f"shader-clock cycles per step: {raw[13] - raw[12]} -> {(raw[13] - raw[12]) / ((t[11] - t[0]) * 10e-3):.0f} MHz effective")
names = ["load params+samples", "tables", "equaliser fwd", "demap+moments+KL", "lag totals + P", "ELBO windows", "E sum, totals",
         "symbol backward", "channel chain", "equaliser bwd", "Adam"]
d = np.diff(t) * 10e-3  # 10 ns ticks -> us
for nm, v in zip(names, d):
    print(f"{nm:24s} {v:8.2f} us")
print(f"{'step':24s} {d.sum():8.2f} us")

sub = raw[14:21]
for nm, v in zip(["pass 1 (u, tt, scalars)", "pass 2 (Hw, D stores, F)", "quad outer products", "barrier", "gather + barrier", "fold + LDS sum"], np.diff(sub) * 10e-3):
    print(f"  phase 0: {nm:26s} {v:6.2f} us")
