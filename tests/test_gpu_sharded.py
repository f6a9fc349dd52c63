"""GPU tests of the halo-sharded step (two processes sharing cuda:0, gloo for the two tiny all-reduces) and of the
generic kernels against the matrix-core kernels."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["V2VAE_ROOT"])
from volterra_vae_amd.distributed import ShardedTrainer, shard_ranges
from volterra_vae_amd.ops import Plan
from volterra_vae_amd.synth import nonlinear_isi_range, pam_constellation, qam_constellation
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
backend = os.environ.get("V2VAE_BACKEND", "gloo")
if backend == "nccl":   # RCCL: the collectives of the step run on the GPU's stream, as on a multi-GPU node
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
L, S, l1, l2, M, B = [int(v) for v in os.environ["V2VAE_SHAPE"].split(",")]
n_glob = B * S
real = bool(os.environ.get("V2VAE_REAL"))   # the real (PAM) model: one stream
ns = 1 if real else 2
sr = shard_ranges(rank, world, n_glob, S, L, l1, l1 // 2, l2)
plan = Plan(ns=ns, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=n_glob, device=dev, y_lo=sr.y_lo, y_len=sr.y_len,
            sym_lo=sr.sym_lo, sym_cnt=sr.sym_cnt, own_lo=sr.own_lo, own_cnt=sr.own_cnt)
rx, _ = nonlinear_isi_range(77, pam_constellation(M) if real else qam_constellation(M * M), S, 16.0, 0, B, B)
y = (np.real(rx)[None, :] if real else np.stack((rx.real, rx.imag)))[:, sr.y_lo:sr.y_lo + sr.y_len]
y = torch.from_numpy(np.ascontiguousarray(y)).float().to(dev)
n = plan.sizes
g = torch.Generator().manual_seed(5)
params = (0.02 * torch.randn(sum(n), generator=g)).to(dev)
params[L // 2] += 1.0; params[n[0] + n[1] + l1 // 2] += 1.0; params[-ns * M:] = 1.0
grads, am, av, avm = (torch.zeros_like(params) for _ in range(4))
lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev); levels = torch.stack([lv] * ns).contiguous()
prior = torch.full_like(levels, 1.0 / M); nv = torch.tensor([0.05], device=dev)
xhat = torch.empty((ns, sr.sym_cnt), device=dev); loss = torch.zeros(1, device=dev)
tr = ShardedTrainer(plan)
kw = dict(y=y, levels=levels, prior=prior, params=params, grads=grads, adam_m=am, adam_v=av, adam_vmax=avm, noise_var=nv,
          xhat=xhat, loss_out=loss, lr=5e-3)
for step in (1, 2):
    if os.environ.get("V2VAE_UNSHARDED"):   # the single-GPU entry point, for comparison
        plan.train_step(step=step, **kw)
    else:
        tr.step(step=step, **kw)
torch.cuda.synchronize()
np.savez(os.environ["V2VAE_OUT"] + f".{rank}.npz", params=params.cpu().numpy(), loss=loss.cpu().numpy(), nv=nv.cpu().numpy(),
         grads=grads.cpu().numpy(),
         xhat=xhat[:, sr.own_lo - sr.sym_lo: sr.own_lo - sr.sym_lo + sr.own_cnt].cpu().numpy(), own_lo=sr.own_lo)
dist.destroy_process_group()
'''


def _run_world(world, tmp_path, shape=(25, 2, 25, 25, 4, 3000), tag="", **extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / f"w{world}{tag}")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   V2VAE_ROOT=ROOT, V2VAE_OUT=out, V2VAE_SHAPE=",".join(str(v) for v in shape), **extra)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    return [np.load(f"{out}.{r}.npz") for r in range(world)]


# (L, S, l1, l2, levels per axis, symbols): the headline shape; the long-channel / long-equaliser 64-QAM shape of the
# 8-GPU configuration (blocked matrix-core kernels with non-zero own_lo / sym_lo / y_lo on rank 1); one sample per symbol
SHARD_SHAPES = [(25, 2, 25, 25, 4, 3000), (65, 2, 65, 65, 8, 3000), (25, 1, 25, 25, 4, 3000), (65, 2, 25, 25, 8, 2500),
                (33, 4, 9, 5, 4, 2600)]   # four samples per symbol: the run ELBO kernel with two waves per phase


@pytest.mark.parametrize("shape", SHARD_SHAPES, ids=lambda s: "L%dS%dl%d_%dM%dB%d" % s)
def test_two_rank_sharded_steps_equal_single_rank(tmp_path, shape):
    one = _run_world(1, tmp_path, shape)[0]
    two = _run_world(2, tmp_path, shape)
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))
    for r in two:  # every rank holds the same parameters after the all-reduced update
        assert rel(r["params"], one["params"]) < 2e-6
        assert abs(float(r["loss"].reshape(-1)[0]) - float(one["loss"].reshape(-1)[0])) < 1e-6 * abs(float(one["loss"].reshape(-1)[0]))
        assert abs(float(r["nv"].reshape(-1)[0]) - float(one["nv"].reshape(-1)[0])) < 1e-6 * float(one["nv"].reshape(-1)[0])
    xs = np.concatenate([r["xhat"] for r in sorted(two, key=lambda r: int(r["own_lo"]))], axis=1)
    assert rel(xs, one["xhat"]) < 1e-6


def test_real_model_on_matrix_cores_sharded_and_against_generic_kernels(tmp_path):
    """The real (PAM) model on the matrix-core kernels (run ELBO kernel with a zero quadrature stream, one-product equaliser
    backward, symmetric-form forward with sums in double): two ranks (non-zero own_lo / sym_lo / y_lo on rank 1) against one, and
    one rank against the generic kernels."""
    shape = (25, 2, 25, 25, 4, 3000)
    one = _run_world(1, tmp_path, shape, tag="r1", V2VAE_REAL="1", V2VAE_REAL_MFMA="1")[0]
    two = _run_world(2, tmp_path, shape, tag="r2", V2VAE_REAL="1", V2VAE_REAL_MFMA="1")
    gen = _run_world(1, tmp_path, shape, tag="rg", V2VAE_REAL="1")[0]   # 3000 symbols without the flag: generic kernels
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64))
    for r in two:
        assert rel(r["params"], one["params"]) < 2e-6
        assert abs(float(r["loss"].reshape(-1)[0]) - float(one["loss"].reshape(-1)[0])) < 1e-6 * abs(float(one["loss"].reshape(-1)[0]))
    xs = np.concatenate([r["xhat"] for r in sorted(two, key=lambda r: int(r["own_lo"]))], axis=1)
    assert rel(xs, one["xhat"]) < 1e-6
    assert rel(one["xhat"], gen["xhat"]) < 1e-6
    assert rel(one["params"], gen["params"]) < 5e-6
    assert abs(float(one["loss"].reshape(-1)[0]) - float(gen["loss"].reshape(-1)[0])) < 1e-6 * abs(float(gen["loss"].reshape(-1)[0]))


def test_sharded_trainer_over_rccl_equals_single_gpu_step_bit_for_bit(tmp_path):
    """ShardedTrainer with the `nccl` backend (RCCL) in a one-rank group on the one GPU: its two all-reduces run through
    RCCL on the compute stream, and with one rank they change nothing, so parameters, gradients, loss and noise variance
    must equal Plan.train_step's exactly (same kernels, same order)."""
    shape = (25, 2, 25, 25, 4, 3000)
    a = _run_world(1, tmp_path, shape, tag="nccl", V2VAE_BACKEND="nccl")[0]
    b = _run_world(1, tmp_path, shape, tag="plain", V2VAE_UNSHARDED="1")[0]
    for k in ("params", "grads", "loss", "nv", "xhat"):
        assert np.array_equal(a[k], b[k]), k


def _step_outputs(env_extra, shape):
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["V2VAE_ROOT"])
sys.path.insert(0, os.path.join(os.environ["V2VAE_ROOT"], "tests"))
from test_gpu_parity import _complex_inputs, run_step, dev
from volterra_vae_amd.ops import Plan
L, S, l1, l2, B, M = [int(v) for v in os.environ["V2VAE_SHAPE"].split(",")]
y, levels, prior, scaling, h1, h2, k1, k2 = _complex_inputs(3, L, S, l1, l2, B, M)
plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
out = run_step(plan, dev(y), dev(h1), dev(h2), dev(k1), dev(k2), dev(levels), dev(prior), dev(scaling), dev(np.array([0.08])))
np.savez(os.environ["V2VAE_OUT"], xhat=out["xhat"].cpu().numpy(), E=out["E"], loss=out["loss"],
         **{"g_" + k: v.cpu().numpy() for k, v in out["g"].items()})
'''
    env = dict(os.environ, V2VAE_ROOT=ROOT, V2VAE_SHAPE=",".join(str(v) for v in shape), **env_extra)
    subprocess.run([sys.executable, "-c", code], env=env, check=True, timeout=240)
    return np.load(env["V2VAE_OUT"])


def test_generic_kernels_agree_with_matrix_core_kernels(tmp_path):
    shape = (25, 2, 25, 25, 5000, 4)
    a = _step_outputs(dict(V2VAE_OUT=str(tmp_path / "mfma.npz")), shape)
    b = _step_outputs(dict(V2VAE_OUT=str(tmp_path / "gen.npz"), V2VAE_FORCE_GENERIC="1"), shape)
    rel = lambda x, y: np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y.astype(np.float64))
    assert abs(float(a["E"]) - float(b["E"])) < 1e-6 * abs(float(b["E"]))
    for k in ("xhat", "g_h1", "g_h2", "g_k1", "g_k2", "g_sc"):
        assert rel(a[k], b[k]) < 1e-5, k
