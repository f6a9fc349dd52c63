"""Timeline of one training step from a rocprofv3 kernel trace: per kernel its duration and the idle gap in front of it, averaged over
the steps of the run.  Usage: python tools/kernel_gaps.py <dir of rocprofv3 --kernel-trace --output-format csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at every k_prep_tables_eq after the warm-up; use the last 20 steps
starts = [i for i, n in enumerate(names) if "k_prep_tables_eq" in n]
starts = starts[-21:]
dur = collections.OrderedDict(); gap = collections.OrderedDict(); cnt = collections.Counter()
tot = 0.0
for a, b in zip(starts[:-1], starts[1:]):
    for i in range(a, b):
        n = names[i].split("(")[0][-48:] + "#" + str(sum(1 for j in range(a, i) if names[j] == names[i]))
        d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        g = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
        dur[n] = dur.get(n, 0) + d; gap[n] = gap.get(n, 0) + g; cnt[n] += 1
    tot += int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
ns = len(starts) - 1
print(f"{ns} steps, {tot / ns / 1e3:.1f} us per step")
sd = sg = 0.0
for n in dur:
    print(f"{n:52s} {dur[n] / ns / 1e3:8.1f} us   gap before {gap[n] / ns / 1e3:6.1f} us")
    sd += dur[n] / ns / 1e3; sg += gap[n] / ns / 1e3
print(f"kernels {sd:.1f} us, gaps {sg:.1f} us")
