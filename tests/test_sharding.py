"""Halo sharding (N > 1 path) on the CPU: range arithmetic against the library's own validation, and
a world_size-2 gloo run in which each rank evaluates only its owned block (with the numpy emulation of
the kernel arithmetic) and the two all-reduces of ShardedTrainer reproduce the unsharded step."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kernel_math as km
from volterra_vae_amd import _lib
from volterra_vae_amd.distributed import shard_ranges


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("S,L,l1,l2,ns", [(2, 15, 25, 25, 1), (2, 25, 25, 25, 2), (1, 7, 5, 3, 1), (3, 9, 7, 0, 2), (2, 65, 65, 65, 2)])
def test_shard_ranges_satisfy_library_halo_contract(world, S, L, l1, l2, ns):
    b_glob = 4096 + 7
    n_glob = b_glob * S
    covered = 0
    for rank in range(world):
        sr = shard_ranges(rank, world, n_glob, S, L, l1, l1 // 2, l2)
        assert sr.own_lo == covered
        covered += sr.own_cnt
        p = _lib.Problem()
        p.ns, p.mode, p.sps, p.l1, p.pad1, p.l2, p.L, p.C = ns, ns - 1, S, l1, l1 // 2, l2, L, 4
        p.n_glob, p.y_lo, p.y_len = n_glob, sr.y_lo, sr.y_len
        p.sym_lo, p.sym_cnt, p.own_lo, p.own_cnt = sr.sym_lo, sr.sym_cnt, sr.own_lo, sr.own_cnt
        assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.OK
        if world > 1 and 0 < rank < world - 1:  # one sample less on either side must be refused
            p.y_lo, p.y_len = sr.y_lo + 1, sr.y_len - 1
            assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.EHALO
            p.y_lo, p.y_len = sr.y_lo, sr.y_len - 1
            assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.EHALO
    assert covered == b_glob


def _case():
    rng = np.random.default_rng(11)
    L, S, l1, l2, B, M = 9, 2, 7, 5, 96, 4
    y = rng.standard_normal((2, B * S))
    lv = np.arange(-(M - 1), M, 2).astype(float)
    levels = np.stack((lv, lv))
    prior = np.ones_like(levels) / M
    scaling = 1.0 + 0.2 * rng.random((2, M))
    h1 = 0.2 * rng.standard_normal((2, L)); h1[0, L // 2] += 1
    h2 = 0.05 * rng.standard_normal((2, L, L))
    k1 = 0.1 * rng.standard_normal((2, l1)); k1[0, l1 // 2] += 1
    k2 = 0.02 * rng.standard_normal((2, l2, l2))
    return dict(L=L, S=S, l1=l1, l2=l2, B=B, y=y, levels=levels, prior=prior, scaling=scaling, h1=h1, h2=h2, k1=k1, k2=k2, nv=0.3)


def _rank_step(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = _case()
    L, S, l1, l2, B = c["L"], c["S"], c["l1"], c["l2"], c["B"]
    N = B * S
    W = N - L + 1
    pre, end = km.delays(L)
    sr = shard_ranges(rank, world, N, S, L, l1, l1 // 2, l2)
    # the rank only ever touches its local sample range: everything outside is overwritten with NaN
    y = np.full_like(c["y"], np.nan)
    y[:, sr.y_lo:sr.y_lo + sr.y_len] = c["y"][:, sr.y_lo:sr.y_lo + sr.y_len]
    ysafe = np.nan_to_num(y, nan=0.0)
    K1s, K2s = km.stacked_equaliser_kernels(1, c["k1"], c["k2"])
    xhat = km.equaliser_forward(ysafe, K1s, K2s, S, l1 // 2, B)
    xhat_local = np.full_like(xhat, np.nan)
    xhat_local[:, sr.sym_lo:sr.sym_lo + sr.sym_cnt] = xhat[:, sr.sym_lo:sr.sym_lo + sr.sym_cnt]
    # symbols computed from samples outside the local range would be NaN had they been used: check none are
    chk = km.equaliser_forward(y, K1s, K2s, S, l1 // 2, B)
    assert np.isfinite(chk[:, sr.sym_lo:sr.sym_lo + sr.sym_cnt]).all()
    xh = np.nan_to_num(xhat_local, nan=0.0)
    q, mom, _ = km.demap_moments(xh, c["levels"], c["prior"], c["scaling"], c["nv"], pre, B + end)
    lo, hi = sr.own_lo, sr.own_lo + sr.own_cnt
    kl = np.sum(q[:, :, max(lo, pre):min(hi, B + end)] * np.log(q[:, :, max(lo, pre):min(hi, B + end)] / c["prior"][:, :, None] + km.EPS))
    tabs = km.polyphase_tables(*km.stacked_channel(1, c["h1"], c["h2"]), L, S)
    E, gmom, parts = km.elbo_fused(ysafe, mom, tabs, L, S, own=(lo, hi))
    scal = torch.tensor([E, kl], dtype=torch.float64)
    dist.all_reduce(scal)                                   # collective 1: [E, KL]
    E_g, kl_g = float(scal[0]), float(scal[1])
    lam = W / E_g
    g_h1, g_h2 = km.finish_channel_grads(1, parts, tabs, L, S, lam)
    gx, gs = km.symbol_backward(xh, c["levels"], c["prior"], c["scaling"], c["nv"], gmom, lam, pre, B + end)
    own_mask = np.zeros(B); own_mask[lo:hi] = 1.0
    gx = gx * own_mask
    dz_scale = own_mask  # scaling gradient only from owned symbols
    _, gs = km.symbol_backward(xh[:, lo:hi], c["levels"], c["prior"], c["scaling"], c["nv"], gmom[:, :, lo:hi], lam,
                               max(pre - lo, 0), min(max(B + end - lo, 0), hi - lo))
    g_k1, g_k2 = km.equaliser_backward(1, ysafe, gx, l1, l2, S, l1 // 2)
    flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in (g_h1, g_h2, g_k1, g_k2, gs)]))
    dist.all_reduce(flat)                                   # collective 2: the flat gradient pack
    if rank == 0:
        out.put((kl_g + W * np.log(E_g), flat.numpy()))
    dist.destroy_process_group()


def test_two_rank_gloo_step_equals_unsharded_step():
    c = _case()
    ref = km.full_step_grads(1, c["y"], (c["h1"], c["h2"], c["k1"], c["k2"]), c["levels"], c["prior"], c["scaling"], c["nv"], c["S"])
    ref_flat = np.concatenate([ref[k].reshape(-1) for k in ("g_h1", "g_h2", "g_k1", "g_k2", "g_scaling")])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_rank_step, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    loss, flat = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(loss - ref["loss"]) < 1e-10 * abs(ref["loss"])
    assert np.linalg.norm(flat - ref_flat) < 1e-9 * np.linalg.norm(ref_flat)


def _rank_small_allreduce(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from volterra_vae_amd.distributed import all_reduce_small, collectives_need_host_staging
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    t = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
    all_reduce_small(t[:2])
    m = torch.tensor([float(rank)])
    all_reduce_small(m, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put((t.tolist(), float(m), collectives_need_host_staging()))
    dist.destroy_process_group()


def test_small_all_reduce_over_gloo_two_ranks():
    """The helper the sharded step and bench.py reduce their payloads with: sum and max over two gloo ranks, and gloo is
    reported as a backend whose device tensors are staged through host memory."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_rank_small_allreduce, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    total, mx, staged = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == [3.0, 30.0] and mx == 1.0 and staged is True
