#!/bin/bash
# tools/dbg: how the waves of each kernel of a loop spend their cycles — issuing / issue-stalled / parked (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY,
# SQ_WAIT_ANY over SQ_WAVE_CYCLES) + MFMA busy.  usage: prof_issue_split.sh TAG [t256|t32|c5]
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}; W=${2:-t256}
O=gpurun_out/issue_${W}_$TAG
rm -rf $O; mkdir -p $O
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
if [ "$W" = t256 ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/p -- python3 bench.py --sde-steps 25 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-roofline > $O/p.log 2>&1
else
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/p -- python3 tools/dbg/${W}_prof.py > $O/p.log 2>&1
fi
python3 tools/dbg/pmc_table.py $O/p > gpurun_out/issue_${W}_$TAG.txt 2>&1
find $O -name "*.csv" -size +2M -delete
python3 - <<PY
for l in open("gpurun_out/issue_${W}_$TAG.txt"):
    if "derived" in l or "wave cycles" in l:
        name = l[:58].strip(); d = l.split("  ")[-1].strip()
        print("%-58s %s" % (name, d))
PY
