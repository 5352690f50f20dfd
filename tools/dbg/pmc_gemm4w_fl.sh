#!/bin/bash
# tools/dbg: SQ / LDS counters of the 4-wave full-line experiment and two of its ablations (one shape), rocprofv3 --pmc only.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
B=$PWD/tools/dbg/build
OUT=$PWD/gpurun_out/pmc_fl
rm -rf $OUT; mkdir -p $OUT
for v in base noread nodma; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS \
     -d $OUT/$v --output-format csv -- $B/gemm4w_fl_$v 16384 1024 4096 0 > $OUT/$v.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for v in ("base", "noread", "nodma"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_fl/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm4w" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
