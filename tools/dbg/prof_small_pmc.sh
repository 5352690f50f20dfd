#!/bin/bash
# tools/dbg: PMC passes over the small-batch sampling loops (shipped T = 32 at B = 64: t32_prof.py; configs[4] share: c5_prof.py): MFMA busy, LDS bank
# conflicts, HBM fetch / write bytes per kernel.  Separate passes, program directly behind `--`.   usage: prof_small_pmc.sh TAG [t32|c5]
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}; W=${2:-t32}
O=gpurun_out/pmc_${W}_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 tools/dbg/${W}_prof.py > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/lds -- python3 tools/dbg/${W}_prof.py > $O/lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/dbg/${W}_prof.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/dbg/${W}_prof.py > $O/write.log 2>&1
python3 tools/dbg/pmc_table.py $O/mfma $O/lds $O/fetch $O/write > gpurun_out/pmc_${W}_$TAG.txt 2>&1
find $O -name "*.csv" -size +2M -delete
head -30 gpurun_out/pmc_${W}_$TAG.txt | cut -c1-400
