#!/bin/bash
# GPU box: the headline bench line with two builds of the library, alternating, on the same box.  Usage: bash tools/ab_bench.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    V2VAE_LIB_PATH=$L timeout -k 10 150 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_tmp.json || exit 1
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_tmp.json"))
print("$L", round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["kernel_ms"].items() if k in ("elbo_fused", "elbo_prepass", "eq_bwd", "eq_fwd")})
PY
  done
done
