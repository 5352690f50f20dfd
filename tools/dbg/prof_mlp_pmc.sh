#!/bin/bash
# tools/dbg: SQ counter passes over the fused MLP kernel alone (mlp_prof.py): where its wave-cycles go (parked / issue-stalled / issuing, per
# instruction class), instruction counts, LDS conflicts.  Separate passes, program directly behind `--`.   usage: prof_mlp_pmc.sh TAG
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}
O=gpurun_out/pmc_mlp_$TAG
rm -rf $O; mkdir -p $O
run() { n=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python3 tools/dbg/mlp_prof.py > $O/$n.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "pass $n killed (rc $rc): stopping"; exit 1; fi; }
run time SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run cls SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
run cnt SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS
python3 tools/dbg/pmc_table.py $O/time $O/cls $O/cnt $O/lds > gpurun_out/pmc_mlp_$TAG.txt 2>&1
find $O -name "*.csv" -size +2M -delete
grep -i "kernel\|ln_mlp" gpurun_out/pmc_mlp_$TAG.txt | cut -c1-1500
tail -3 $O/*.log | cut -c1-300
