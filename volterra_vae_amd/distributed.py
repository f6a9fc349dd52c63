"""Halo sharding of one long batch over the GPUs of a node (one process per GPU, RCCL).

The ELBO, the KL term and every gradient are sums over windows / symbols, so one batch splits
into contiguous symbol blocks.  Rank r owns symbols [own_lo, own_lo + own_cnt): it evaluates their
KL and per-symbol gradients, the windows n in [own_lo*S, (own_lo+own_cnt)*S) and the equaliser
gradients of its symbols.  To do that without any exchange of activations it holds
  * moments for `kmax = (S-1+L-1)//S` extra symbols on each side (recomputed locally), and
  * received samples for those symbols plus the equaliser's lag halo,
with zero padding only at the two global ends, i.e. exactly the un-sharded single big batch.
Per step there are two collectives, both tiny: the scalars [E, KL] (2 doubles) and the flat
gradient pack, between three calls into the library (v2vae_train_step_sharded_pre / _post, v2vae_train_step_update).  The reference has no distributed code (its `fit` is a sequential mini-batch loop,
lib/real_equalization.py:484-502); this module is new.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardRanges:
    own_lo: int
    own_cnt: int
    sym_lo: int
    sym_cnt: int
    y_lo: int
    y_len: int


def shard_ranges(rank: int, world: int, n_glob: int, sps: int, L: int, l1: int, pad1: int, l2: int) -> ShardRanges:
    """Contiguous symbol block of `rank` and the symbol / sample halos it needs."""
    if n_glob % sps:
        raise ValueError("n_glob must be a multiple of sps")
    b_glob = n_glob // sps
    if b_glob < world:
        raise ValueError("fewer symbols than ranks")
    base, extra = divmod(b_glob, world)
    own_lo = rank * base + min(rank, extra)
    own_cnt = base + (1 if rank < extra else 0)
    kmax = (sps - 1 + L - 1) // sps
    pre = (L - 1) // 2
    p2 = l2 // 2
    sym_lo = max(0, own_lo - kmax)
    sym_hi = min(b_glob, own_lo + own_cnt + kmax)
    left = max(pad1, p2)
    right = max(l1 - 1 - pad1, p2)
    y_lo = min(sym_lo * sps - left, (own_lo - kmax) * sps + pre)
    y_hi = max((sym_hi - 1) * sps + right + 1, (own_lo + own_cnt) * sps + pre)
    y_lo = max(0, y_lo)
    y_hi = min(n_glob, y_hi)
    return ShardRanges(own_lo, own_cnt, sym_lo, sym_hi - sym_lo, y_lo, y_hi - y_lo)


def collectives_need_host_staging(group=None) -> bool:
    """True when the group's backend has no device collectives (gloo: rehearsals of the multi-rank path with several
    ranks on one GPU, CPU tests).  RCCL ("nccl") reduces device tensors on the compute stream and needs none."""
    import torch.distributed as dist
    return dist.get_backend(group) != "nccl"


def all_reduce_small(t, group=None, op=None, host_staged=None):
    """In-place all-reduce of a small tensor (the step's payloads are 2 doubles and the <= 34 KB gradient pack).

    Over RCCL the device tensor is reduced as it is.  Over gloo a device tensor goes through host memory explicitly: copy to
    the host (which waits for the kernels queued before it), gloo on host memory, copy back.  gloo's own handling of device
    tensors (a pinned staging copy on a side stream that waits for an event of the compute stream, synchronised from a worker
    thread) stalls intermittently on this stack when about a millisecond of kernels is queued in front of it and two
    processes share the GPU - with torch kernels alone as well as with this library's (tools/gloo_cuda_repro.py,
    tools/rehearse_shard.py) - so it is not used."""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if host_staged is None:
        host_staged = collectives_need_host_staging(group)
    if host_staged and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


class ShardedTrainer:
    """One data-parallel V2VAE training step over a halo-sharded batch.

    Every rank holds identical parameters and Adam state; after the two all-reduces every rank
    applies the identical update, so no parameter broadcast is needed after initialisation."""

    def __init__(self, plan, process_group=None):
        import torch.distributed as dist
        self.plan = plan
        self.dist = dist
        self.group = process_group
        self.host_staged = collectives_need_host_staging(process_group)

    def step(self, *, accumulate_scaling_grad=False, **kw):
        """Keyword arguments as for Plan.train_args (y, levels, prior, params, grads, adam_m, adam_v, adam_vmax, noise_var,
        xhat, loss_out, lr, step, learn_scaling, adaptive_noise, freeze_h2, betas, eps).  Three library calls and two
        all-reduces; nothing else runs on the host between them."""
        if accumulate_scaling_grad:
            raise NotImplementedError("the halo-sharded step does not keep a running noise-scaling gradient "
                                      "(memory-polynomial / Hammerstein classes): their reference semantics are per mini-batch")
        plan, dist = self.plan, self.dist
        a = plan.train_args(**kw)
        plan.sharded_pre(a)
        all_reduce_small(plan.scal[:2], self.group, host_staged=self.host_staged)
        plan.sharded_post(a)
        all_reduce_small(kw["grads"], self.group, host_staged=self.host_staged)
        plan.update(a)
