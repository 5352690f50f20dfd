#!/bin/bash
# tools/dbg: kernel-trace statistics of the two small-M sampling regimes (shipped T = 32 at B = 64; configs[4] share B = 32 conditioned)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}
for w in t32 c5; do
  rm -rf gpurun_out/prof_${w}_$TAG
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${w}_$TAG --output-format csv -- python3 tools/dbg/${w}_prof.py > gpurun_out/prof_${w}_$TAG.log 2>&1
  f=$(find gpurun_out/prof_${w}_$TAG -name "*kernel_stats.csv" | head -1)
  echo "== $w ($f)"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over 120 steps = %.3f ms/step" % (tot / 1e6, tot / 1e6 / 120))
for r in rows[:14]:
    print("%6.1f%%  calls %6s  avg %8.1f us  %s" % (float(r["Percentage"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
done
