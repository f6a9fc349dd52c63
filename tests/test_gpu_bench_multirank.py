"""bench.py launched the way the scaling run launches it (python -m torch.distributed.run, one process per rank), with two
ranks sharing the one GPU of the test box and gloo standing in for RCCL (RCCL refuses two ranks on one device).  Each case
must print exactly one JSON line with n_gpus == 2 and a finite loss.  The halo-sharded step has no counterpart in the
reference (lib/real_equalization.py:484-502 is a sequential loop)."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun_bench(extra, world=2, timeout=400):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, V2VAE_BENCH_BACKEND="gloo", V2VAE_BENCH_WATCHDOG="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", str(world)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["unit"] == "symbols/s" and out["value"] > 0
    return out


@pytest.mark.parametrize("log2_symbols,steps,warmup", [(16, 3, 1), (22, 12, 3)], ids=["2p16", "2p22_headline_size"])
def test_two_rank_headline_bench_completes(log2_symbols, steps, warmup):
    """The 2^22-symbol case is the configuration whose first all-reduce never returned while gloo handled the device tensors
    itself (about a millisecond of kernels queued in front of the collective)."""
    out = _torchrun_bench(["--steps", str(steps), "--warmup", str(warmup), "--log2-symbols", str(log2_symbols), "--no-cpu-baseline"])
    assert out["steps"] == steps and out["scaling"] == "weak"
    assert out["config"]["symbols_per_gpu"] == 1 << log2_symbols
    assert math.isfinite(out["final_loss"])
    assert out["roofline"]["kernel_ms"] > 0


def test_two_rank_long_channel_bench_completes():
    out = _torchrun_bench(["--workload", "cfg4", "--steps", "3", "--warmup", "1", "--log2-symbols", "13", "--no-cpu-baseline"])
    assert "L=65" in out["config"]["workload"] and "64-QAM" in out["config"]["workload"]
    assert math.isfinite(out["final_loss"])


def test_two_rank_sweep_bench_completes():
    out = _torchrun_bench(["--workload", "sweep", "--steps", "3", "--warmup", "1"])
    assert out["config"]["models_per_gpu"] == 512 and out["scaling"] == "strong"
    assert math.isfinite(out["final_loss_mean"])
