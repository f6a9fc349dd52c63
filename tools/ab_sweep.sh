#!/bin/bash
# GPU box: the sweep bench line (1024 models, on-chip fit kernel) with two builds of the library, alternating.  Usage: bash tools/ab_sweep.sh <libA> <libB>
for rep in 1 2; do
  for L in $1 $2; do
    V2VAE_LIB_PATH=$L timeout -k 10 200 python bench.py --workload sweep --no-cpu-baseline > gpurun_out/sw_tmp.json || exit 1
    python - <<PY
import json
d = json.load(open("gpurun_out/sw_tmp.json"))
print("$L", round(d["ms_per_step"], 4), round(d["value"] / 1e9, 3))
PY
  done
done
