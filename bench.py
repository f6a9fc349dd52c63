#!/usr/bin/env python
"""V2VAE train-step throughput on MI355X.

A step = forward + ELBO + backward + Adam-AMSGrad update of one batch, inputs resident in HBM.
Workload at 1 GPU: complex 16-QAM V2VAE, channel memory L=25, equaliser l1=l2=25, 2 samples per
symbol, one batch of 2^22 symbols (BASELINE.json configs[1]; S, l1, l2 and "one batch = the whole
block" are this build's stated assumptions, the baseline file does not fix them).  With N GPUs the
batch grows to N * 2^22 symbols, halo-sharded by contiguous blocks (weak scaling), with two tiny
RCCL all-reduces per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X dense fp32 (vector == f32 MFMA) peak


def algorithmic_flops_per_symbol(L, S, l1, l2, cplx):
    """Per-symbol algorithmic flops (2 per multiply-add) of each compute kernel.

    Equaliser (real): forward 2 l1 + 2 l2^2 + 2 l2, backward 2 l1 + 2 l2^2; complex arithmetic costs 4 real
    multiply-adds per complex one.  ELBO per window on Le = L / S active taps, S windows per symbol:
      real     H mu, (H.H) v, H w (3 mat-vecs) + 3 rank-1 accumulations           = 12 Le^2 flops
      complex  H X and H w (complex mat-vecs, 8 Le^2 each), one complex rank-1 accumulation (8 Le^2),
               = 24 Le^2 flops; the s s^T accumulation and the (|H|^2) s term of E with its gradient are O(Le) per
               symbol (lagged products, correlations) and not counted
    i.e. only work the kernels actually execute is counted (the stacked real-valued formulation would be 48 Le^2)."""
    c = 4 if cplx else 1
    return {
        # complex, up to 95 lags (k_eq_fwd_sym): symmetric form, l2 (l2 + 1) / 2 + l2 complex multiply-adds (first-order kernel
        # folded in); beyond that the full lag matrix is evaluated
        "eq_fwd": 8 * (l2 * (l2 + 1) // 2 + l2) if (cplx and l2 <= 95) else c * (2 * l1 + 2 * l2 * l2 + 2 * l2),
        # the gradient planes are symmetric in the lag pair: the same count (k_eq_bwd_bf, up to 95 lags, accumulates the upper-
        # triangular 16x16 blocks only)
        "eq_bwd": 8 * (l2 * (l2 + 1) // 2 + l2) if (cplx and l2 <= 95) else c * (2 * l1 + 2 * l2 * l2),
        "elbo_fused": (24 if cplx else 12) * L * L / S,
    }


def cpu_baseline(L, S, l1, l2, order, budget_s=12.0):
    import torch
    from oracle import v2vae_oracle as orc
    from volterra_vae_amd.synth import nonlinear_isi_range, qam_constellation
    nthreads = torch.get_num_threads()
    B = 1 << 15
    rx, _ = nonlinear_isi_range(1234, qam_constellation(order), S, 16.0, 0, B, B)
    y = torch.from_numpy(np.stack((rx.real, rx.imag))).to(torch.float32)
    port = orc.ComplexV2VAEPort(L, l1, l2, order, S, 5e-3, 10 ** -1.6, dtype=torch.float32)
    port.step(y)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        port.step(y)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": n * B / el, "unit": "symbols/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} train steps of one {B}-symbol batch (same model shape, fp32, torch CPU oracle, window form)"}


def sweep_main(args):
    """BASELINE.json configs[4]: 1024 independent channel/SNR points, real PAM-4, L=15 (S=2, l1=l2=25 and 500-symbol
    batches as in the reference's default script), models split over the ranks, no collective.  A step = one
    mini-batch training step of every model; the timed region is ONE launch of --steps batches per rank."""
    import torch
    import torch.distributed as dist
    from volterra_vae_amd.distributed import all_reduce_small
    from volterra_vae_amd.sweep import SecondVolterraV2VAESweep, shard_models
    from volterra_vae_amd.synth import nonlinear_isi, pam_constellation
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    backend = os.environ.get("V2VAE_BENCH_BACKEND", "nccl")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_total, L, S, l1, l2, B, C = 1024, 15, 2, 25, 25, 500, 4
    lo, hi = shard_models(rank, world, n_total)
    M = hi - lo
    const = pam_constellation(C)
    nbt = args.warmup + args.steps
    # 16 distinct synthetic sequences (SNR 10..25 dB) shared round-robin by the models of this rank
    snrs = np.linspace(10.0, 25.0, 16)
    base = np.stack([nonlinear_isi(nbt * B, const, S, snr, np.random.default_rng(100 + i))[0] for i, snr in enumerate(snrs)])
    idx = (np.arange(lo, hi) % 16)
    sweep = SecondVolterraV2VAESweep(M, L, l1, l2, 5e-3, const, S, B, noise_variance=10 ** (-snrs[idx] / 10), torch_device=dev,
                                     lr_schedule=None)
    y = torch.from_numpy(base[idx].astype(np.float32)).view(M, 1, -1).to(dev).contiguous()
    lv, prior = sweep.constellation.view(1, -1), sweep.constellation_prior.view(1, -1)
    kw = dict(levels=lv, prior=prior, params=sweep.params, grads=sweep._grads, adam_m=sweep._m, adam_v=sweep._v,
              adam_vmax=sweep._vmax, noise_var=sweep.noise_variance)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup:
        sweep._fit.run(y=y[:, :, :args.warmup * B * S].contiguous(), lrs=[5e-3] * args.warmup, step0=0, **kw)
    ys = y[:, :, args.warmup * B * S:].contiguous()
    lc = torch.zeros((M, args.steps), dtype=torch.float32, device=dev)
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    sweep._fit.run(y=ys, lrs=[5e-3] * args.steps, step0=args.warmup, loss_curve=lc, **kw)
    e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_small(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if not bool(torch.isfinite(lc).all()):
        raise SystemExit("non-finite loss in the sweep")
    if rank == 0:
        fl = algorithmic_flops_per_symbol(L, S, l1, l2, False)
        flops_launch = sum(fl.values()) * M * B * args.steps
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "V2VAE train-step symbols/sec (fwd+bwd) at 1/2/4/8 MI355X; % of HBM/MFMA roofline",
            "value": n_total * B * args.steps / elapsed, "unit": "symbols/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batched sweep: {n_total} independent real PAM-4 V2VAE models, L=15, l1=l2=25, sps=2, "
                                   f"{B}-symbol mini-batches, Adam-AMSGrad, one persistent launch per rank",
                       "models_per_gpu": M, "parallelism": f"models split over {world} ranks, no collective"},
            "roofline": {"bound": "mfma", "kernel": "k_fit_rs", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                         "algorithmic_flops_per_symbol": sum(fl.values()), "kernel_ms": kernel_ms},
            "final_loss_mean": float(lc[:, -1].mean()),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--log2-symbols", type=int, default=None,
                    help="symbols per GPU = 2^this (default 22; 23 with --workload cfg4; ignored by --workload sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--channel-memory", type=int, default=25, help="L (65 with --order 64 = BASELINE.json configs[3] per GPU)")
    ap.add_argument("--lags", type=int, default=25, help="equaliser lags l1 = l2")
    ap.add_argument("--order", type=int, default=16, help="QAM order")
    ap.add_argument("--data-source", choices=["host", "device"], default="host",
                    help="host: numpy generator + one H2D copy (default, as in earlier rounds); device: the sequence is produced "
                         "in HBM by v2vae_synth_channel (no host work; meant for the 2^26-symbol configuration)")
    ap.add_argument("--workload", choices=["train_step", "cfg4", "sweep"], default="train_step",
                    help="train_step: the headline configuration, BASELINE.json configs[1] per GPU (default, at every --gpus N, so "
                         "that the per-N values form one weak-scaling curve); cfg4: the shape of BASELINE.json configs[3] per GPU "
                         "(complex 64-QAM, L=65, l1=l2=65, 2^23 symbols per GPU = 2^26 over 8, data generated on the device); "
                         "sweep: 1024 independent real PAM-4 models, L=15")
    args = ap.parse_args()
    if os.environ.get("V2VAE_BENCH_WATCHDOG"):   # diagnostic: every thread's Python stack after that many seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["V2VAE_BENCH_WATCHDOG"]), exit=True)
    if args.workload == "sweep":
        return sweep_main(args)
    if args.workload == "cfg4":
        args.channel_memory, args.lags, args.order, args.data_source = 65, 65, 64, "device"
    if args.log2_symbols is None:
        args.log2_symbols = 23 if args.workload == "cfg4" else 22

    import torch
    import torch.distributed as dist
    from volterra_vae_amd.distributed import ShardedTrainer, all_reduce_small, shard_ranges
    from volterra_vae_amd.ops import Plan
    from volterra_vae_amd.synth import nonlinear_isi_range, qam_constellation

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    backend = os.environ.get("V2VAE_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the multi-rank path on a one-GPU box
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    L, S, l1, l2, order = args.channel_memory, 2, args.lags, args.lags, args.order
    M = int(round(np.sqrt(order)))
    b_per_gpu = 1 << args.log2_symbols
    b_glob = b_per_gpu * world
    n_glob = b_glob * S
    sr = shard_ranges(rank, world, n_glob, S, L, l1, l1 // 2, l2)
    plan = Plan(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=n_glob, device=dev,
                y_lo=sr.y_lo, y_len=sr.y_len, sym_lo=sr.sym_lo, sym_cnt=sr.sym_cnt, own_lo=sr.own_lo, own_cnt=sr.own_cnt)

    # synthetic nonlinear-ISI input for this rank's sample range (identical across ranks where ranges overlap)
    assert sr.y_lo % S == 0 or True
    s_lo, s_hi = sr.y_lo // S, -(-(sr.y_lo + sr.y_len) // S)
    if args.data_source == "device":
        from volterra_vae_amd.synth_device import nonlinear_isi_device
        m_side = int(round(np.sqrt(order)))
        yy = nonlinear_isi_device(20250321, np.arange(-(m_side - 1), m_side, 2), 2, S, 16.0, b_glob, s_lo, s_hi - s_lo, device=dev)
        y = yy[:, sr.y_lo - s_lo * S: sr.y_lo - s_lo * S + sr.y_len].contiguous()
    else:
        rx, _ = nonlinear_isi_range(20250321, qam_constellation(order), S, 16.0, s_lo, s_hi, b_glob)
        rx = rx[sr.y_lo - s_lo * S: sr.y_lo - s_lo * S + sr.y_len]
        y = torch.from_numpy(np.stack((rx.real, rx.imag))).to(torch.float32).to(dev).contiguous()

    n = plan.sizes
    params = torch.zeros(sum(n), dtype=torch.float32, device=dev)
    params[L // 2] = 1.0                              # h1 = dirac (Re plane)
    params[n[0] + n[1] + l1 // 2] = 1.0               # k1 = dirac (Re plane)
    params[n[0] + n[1] + n[2] + n[3]:] = 1.0          # noise scaling
    grads, am, av, avm = (torch.zeros_like(params) for _ in range(4))
    lv = torch.arange(-(M - 1), M, 2, dtype=torch.float32, device=dev)
    levels = torch.stack((lv, lv)).contiguous()
    prior = torch.full_like(levels, 1.0 / M)
    noise_var = torch.tensor([10 ** -1.6], dtype=torch.float32, device=dev)
    xhat = torch.empty((2, sr.sym_cnt), dtype=torch.float32, device=dev)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    trainer = ShardedTrainer(plan) if world > 1 else None
    kw = dict(y=y, levels=levels, prior=prior, params=params, grads=grads, adam_m=am, adam_v=av, adam_vmax=avm,
              noise_var=noise_var, xhat=xhat, loss_out=loss, lr=5e-3)

    def one_step(i):
        if trainer is None:
            plan.train_step(step=i, **kw)
        else:
            trainer.step(step=i, **kw)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i + 1)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i + 1)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_small(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)
    if not np.isfinite(final_loss):
        raise SystemExit(f"non-finite loss {final_loss}")

    # per-kernel durations with events on the launch stream (torch's current stream), stage by stage
    o = np.cumsum([0] + n)
    h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
    stages = {
        "prep_tables": lambda: plan.prep_tables(h1, h2, k1, k2),
        "eq_fwd": lambda: plan.eq_fwd(y, xhat),
        "demap_moments": lambda: plan.demap_moments(xhat, levels, prior, sc, noise_var),
        "elbo_prepass": lambda: plan.elbo_prepass(),
        "elbo_fused": lambda: plan.elbo_fused(y, prepass=False),
        "reduce_scalars": lambda: (plan.reduce_scalars(), plan.finalize_scalars()),
        "sym_bwd": lambda: plan.sym_bwd(xhat, levels, prior, sc, noise_var),
        "eq_bwd": lambda: plan.eq_bwd(y, plan.gxhat),
        "finish_grads": lambda: plan.finish_grads(7, grads[o[0]:o[1]], grads[o[1]:o[2]], grads[o[2]:o[3]], grads[o[3]:o[4]], grads[o[4]:o[5]]),
    }
    kernel_ms = {k: 0.0 for k in stages}
    reps = max(3, min(args.steps, 10))
    for _ in range(reps):
        for name, fn in stages.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            kernel_ms[name] += e0.elapsed_time(e1) / reps
    fence()

    if rank == 0:
        fl = algorithmic_flops_per_symbol(L, S, l1, l2, True)
        dom = max(fl, key=lambda k: kernel_ms[k])
        achieved = fl[dom] * sr.own_cnt / (kernel_ms[dom] * 1e-3) / 1e12
        step_flops = sum(fl.values()) * b_glob
        ms_per_step = elapsed / args.steps * 1e3
        traffic = None
        try:  # HBM bytes per launch of the dominant kernel, measured with rocprofv3 --pmc passes (committed under profiles/)
            with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as fh:
                tr = json.load(fh)
            if dom == "elbo_fused" and world == 1 and args.log2_symbols == 22 and (L, l1, order) == (25, 25, 16):
                traffic = next(v for k, v in tr.items() if "k_elbo_cx" in k)
        except Exception:
            traffic = None
        # how busy a SIMD's two pipes were in the dominant kernel: (matrix cycles + vector cycles) / elapsed cycles, from the
        # committed SQ counter passes of the same command (tools/profile_step.sh, tools/pipe_bound.py).  With f32 matrix
        # instructions the two shared one issue slot and the sum was a bound; the bf16 ones of the split arithmetic run beside
        # the vector instructions, so the sum can exceed 1 (it does in the equaliser backward)
        pipe = None
        try:
            with open(os.path.join(ROOT, "profiles", "latest_pipe.json")) as fh:
                pj = json.load(fh)
            if dom == "elbo_fused" and world == 1 and args.log2_symbols == 22 and (L, l1, order) == (25, 25, 16):
                pipe = next(v["pipe_bound_frac"] for k, v in pj.items() if "k_elbo_cx" in k)
        except Exception:
            pipe = None
        out = {
            "metric": "V2VAE train-step symbols/sec (fwd+bwd) at 1/2/4/8 MI355X; % of HBM/MFMA roofline",
            "value": b_glob * args.steps / elapsed, "unit": "symbols/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (generated on device)" if args.data_source == "device" else ""),
            "config": {"workload": f"complex {order}-QAM V2VAE train step, L={L}, l1=l2={l1}, sps=2, one batch of {world}x2^{args.log2_symbols} symbols, "
                                   "Adam-AMSGrad, random-init (dirac) kernels, synthetic Wiener-Hammerstein ISI at 16 dB",
                       "symbols_per_gpu": b_per_gpu, "parallelism": f"halo-sharded dp{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "pipe_bound_frac": pipe,
                         "algorithmic_flops_per_symbol": fl[dom], "kernel_ms": kernel_ms[dom],
                         "whole_step_frac": step_flops / world / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                         "arithmetic": "float results; the table products of the matrix-core kernels run on bf16 matrix instructions in "
                                       "three-piece split arithmetic (six piece products, float accumulation), priced against the f32 peak"},
            "kernel_ms": kernel_ms, "final_loss": final_loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, S, l1, l2, order)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
