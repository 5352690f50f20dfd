"""tools/dbg: per-position kernel durations and inter-kernel gaps from a rocprofv3 kernel_trace.csv (steady-state steps of a sampling loop).
usage: python3 tools/dbg/trace_seq.py <kernel_trace.csv> [period_anchor_substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
anchor = sys.argv[2] if len(sys.argv) > 2 else "sampler_step_kernel"
idx = [i for i, n in enumerate(names) if anchor in n]
# steady state: periods between consecutive anchors with the most common length
per = collections.Counter(b - a for a, b in zip(idx, idx[1:])).most_common(1)[0][0]
starts = [a for a, b in zip(idx, idx[1:]) if b - a == per][5:]          # skip warm-up periods
print("period = %d kernels, %d steady periods" % (per, len(starts)))
dur = [0.0] * per; gap = [0.0] * per
for a in starts:
    for j in range(per):
        dur[j] += en[a + j] - st[a + j]
        gap[j] += st[a + j + 1] - en[a + j]
n = len(starts)
tot_d = sum(dur) / n / 1e3; tot_g = sum(gap) / n / 1e3
print("per step: kernels %.1f us + gaps %.1f us = %.1f us" % (tot_d, tot_g, tot_d + tot_g))
# aggregate by (name, position within its block): show first 2 blocks' worth in order, then name totals
a = starts[0]
for j in range(min(per, 22)):
    print("%3d  %8.2f us  gap %6.2f  %s" % (j, dur[j] / n / 1e3, gap[j] / n / 1e3, names[a + j][:90]))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for j in range(per):
    k = agg[names[a + j][:90]]; k[0] += 1; k[1] += dur[j] / n / 1e3; k[2] += gap[j] / n / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x  avg %7.2f us  gap-after avg %5.2f  %s" % (v[0], v[1] / v[0], v[2] / v[0], k))
