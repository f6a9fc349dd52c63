#!/bin/bash
# GPU box: parity of the matrix-core kernels, then the config-4 shape bench line (per-kernel times).  Usage: bash tools/quick_cfg4.sh <tag>
set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/$1_cfg4_bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/$1_cfg4_bench.json"))
print(round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items()})
PY
