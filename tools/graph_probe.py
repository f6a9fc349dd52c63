"""GPU box: what would capturing the training step in a HIP graph gain?  (Timing probe only: the captured step freezes Adam's
step count, so this is not a valid way to train as it stands.)"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from volterra_vae_amd.ops import Plan
from volterra_vae_amd.synth import nonlinear_isi_range, qam_constellation
L, S, l1, l2, M = 25, 2, 25, 25, 4
B = 1 << 22
plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
rx, _ = nonlinear_isi_range(1, qam_constellation(16), S, 16.0, 0, B, B)
y = torch.from_numpy(np.stack((rx.real, rx.imag))).float().cuda().contiguous()
n = plan.sizes
params = torch.zeros(sum(n), device="cuda"); params[L // 2] = 1; params[n[0] + n[1] + l1 // 2] = 1; params[-2 * M:] = 1
grads, am, av, avm = (torch.zeros_like(params) for _ in range(4))
lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device="cuda"); levels = torch.stack((lv, lv)).contiguous()
prior = torch.full_like(levels, 1.0 / M); nv = torch.tensor([10 ** -1.6], device="cuda")
xhat = torch.empty((2, B), device="cuda"); loss = torch.zeros(1, device="cuda")
kw = dict(y=y, levels=levels, prior=prior, params=params, grads=grads, adam_m=am, adam_v=av, adam_vmax=avm, noise_var=nv, xhat=xhat,
          loss_out=loss, lr=5e-3)
for i in range(20): plan.train_step(step=i + 1, **kw)
torch.cuda.synchronize()
def timeit(f, nrep=400):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(nrep): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / nrep * 1e3
print("eager  %.4f ms" % timeit(lambda i: plan.train_step(step=21 + i, **kw)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    plan.train_step(step=500, **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.train_step(step=501, **kw)
torch.cuda.synchronize()
print("graph  %.4f ms" % timeit(lambda i: g.replay()))
print("eager  %.4f ms" % timeit(lambda i: plan.train_step(step=1000 + i, **kw)))
