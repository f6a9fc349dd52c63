import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
import volterra_vae_amd._lib as L
# swap in the instrumented library
lib = ctypes.CDLL(os.path.join(os.getcwd(), ""+os.environ.get("STAMPLIB","tools/libstamps_NONE.so")+""))
for name, (res, args) in L.EXPORTS.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
L.lib = lib
import volterra_vae_amd.ops as ops
ops.lib = lib
from volterra_vae_amd.ops import Plan
from volterra_vae_amd.synth import nonlinear_isi_range, qam_constellation
Lc, S, l1, l2, M = int(os.environ.get("LC", "25")), 2, 25, 25, 4
B = 1 << int(os.environ.get("LOG2B", "22"))
plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=Lc, C_levels=M, n_glob=B * S, device="cuda")
rx, _ = nonlinear_isi_range(1, qam_constellation(16), S, 16.0, 0, B, B)
y = torch.from_numpy(np.stack((rx.real, rx.imag))).float().cuda().contiguous()
n = plan.sizes
params = torch.zeros(sum(n), device="cuda"); params[Lc // 2] = 1; params[n[0] + n[1] + l1 // 2] = 1; params[-2 * M:] = 1
params[n[0]:n[0]+n[1]] = 0.01 * torch.randn(n[1], device="cuda")
o = np.cumsum([0] + n)
h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device="cuda"); levels = torch.stack((lv, lv)).contiguous()
prior = torch.full_like(levels, 1.0 / M); nv = torch.tensor([0.03], device="cuda")
xhat = torch.empty((2, B), device="cuda")
plan.prep_tables(h1, h2, k1, k2); plan.eq_fwd(y, xhat); plan.demap_moments(xhat, levels, prior, sc, nv)
plan.elbo_prepass()
for _ in range(2): plan.elbo_fused(y, prepass=False)
torch.cuda.synchronize()
off = ctypes.CDLL(None)  # dummy
# locate o_sumstage via a second plan-less trick: the debug area is at ws + o_sumstage; find by scanning is hard -> expose through env
import re
lib.v2vae_dbg_offset.restype = ctypes.c_int64
lay_off = lib.v2vae_dbg_offset(ctypes.byref(plan.p))
ws = plan.ws
nb = 256; NW = int(os.environ.get("NW", "4"))
raw = ws[lay_off:lay_off + nb * NW * 8 * 2].cpu().numpy().view(np.uint64).reshape(nb, NW, 8)
tot = raw.sum(axis=2, keepdims=True).astype(np.float64)
np.set_printoptions(linewidth=200, suppress=True)
print("mean cycles per wave per phase:", raw.mean(axis=(0, 1)).round(0))
print("share:", (raw / tot).mean(axis=(0, 1)))
print("total per wave (cycles):", tot.mean())
