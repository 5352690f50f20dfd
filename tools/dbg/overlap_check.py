"""tools/dbg: from a rocprofv3 kernel_trace.csv, how much of each launch of kernel <substr> overlaps other kernels (side-stream concurrency check)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); sub = sys.argv[2]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
tot = ov = n = 0
for i, (s, e, name) in enumerate(ev):
    if sub not in name: continue
    n += 1; tot += e - s
    for j in range(max(0, i - 40), min(len(ev), i + 400)):
        if j == i: continue
        s2, e2, _ = ev[j]
        if s2 >= e: break
        ov += max(0, min(e, e2) - max(s, s2))
print("%d launches of %s: avg %.1f us, overlapped by other kernels for %.1f us on average (sum over overlapping kernels)" % (n, sub, tot / n / 1e3, ov / n / 1e3))
