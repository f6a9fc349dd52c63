#!/usr/bin/env python
"""How busy the one fp32 pipe of a SIMD (matrix and vector instructions share it) was, per kernel, from two rocprofv3 --pmc passes:
    python tools/pipe_bound.py <sq1 dir> <sq2 dir> <out.json>
sq1 holds SQ_VALU_MFMA_BUSY_CYCLES, sq2 holds SQ_INSTS_VALU and GRBM_GUI_ACTIVE (tools/profile_step.sh collects both).
Per launch:  matrix cycles per SIMD = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs,  vector cycles per SIMD = 4 x SQ_INSTS_VALU / 1024
(a wave-wide vector instruction issues over four cycles),  elapsed cycles = GRBM_GUI_ACTIVE / 8 XCDs;
pipe_bound_frac = (matrix + vector) / elapsed."""
import collections
import csv
import glob
import json
import sys


def means(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in acc.items()}


def main():
    a, b = means(sys.argv[1]), means(sys.argv[2])
    out = {}
    for k in a:
        if k not in b or "GRBM_GUI_ACTIVE" not in b[k] or not k.startswith(("v2vae", "k_")):
            continue
        mfma = a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024
        valu = 4 * b[k].get("SQ_INSTS_VALU", 0.0) / 1024
        el = b[k]["GRBM_GUI_ACTIVE"] / 8
        out[k] = {"matrix_cycles_per_simd": mfma, "vector_cycles_per_simd": valu, "elapsed_cycles": el,
                  "pipe_bound_frac": (mfma + valu) / el if el else None,
                  "insts_valu": b[k].get("SQ_INSTS_VALU"), "insts_mfma": b[k].get("SQ_INSTS_MFMA"),
                  "coexec_cycles": b[k].get("SQ_VALU_MFMA_COEXEC_CYCLES")}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["elapsed_cycles"])[:8]:
        print(f"{k[:60]:60s} matrix {v['matrix_cycles_per_simd']:10.0f}  vector {v['vector_cycles_per_simd']:10.0f}  "
              f"elapsed {v['elapsed_cycles']:10.0f}  pipe busy {v['pipe_bound_frac']:.3f}")


if __name__ == "__main__":
    main()
