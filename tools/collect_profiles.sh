#!/bin/bash
# Collect the judged profile artefacts of a round on the GPU box (run through gpurun; results land in gpurun_out/, copy into profiles/):
#   tools/collect_profiles.sh r03
# 1. --kernel-trace --stats of 30 SDE steps of the bench workload; 2./3. PMC FETCH_SIZE / WRITE_SIZE (separate passes); 4. PMC MFMA utilisation.
# The program itself follows `--` (no env / bash wrappers: the profiler's preloaded library initialises the GPU first).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-rXX}
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-extras --no-roofline"
# the two traffic passes keep the roofline pass in: it launches the stand-alone self-attention kernel (`roofline_attention`: in the forward that
# loop is the epilogue of the QKV GEMM and has no launch of its own), so traffic.json gets its FETCH / WRITE bytes too
PMCRUN="--no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --sde-steps 30 --steps 1 --warmup 1 $COMMON > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --sde-steps 25 --steps 1 --warmup 0 $PMCRUN > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --sde-steps 25 --steps 1 --warmup 0 $PMCRUN > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 bench.py --sde-steps 25 --steps 1 --warmup 0 $COMMON > $O/mfma.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_sde30.csv
python3 tools/reduce_pmc.py $TAG $O/fetch $O/write "bench.py --sde-steps 25 (B=64, T=256), rocprofv3 --pmc, $TAG final kernels" | tail -2
python3 tools/reduce_pmc_mfma.py $TAG $O/mfma "bench.py --sde-steps 25 (B=64, T=256), rocprofv3 --pmc, $TAG final kernels"
cp profiles/${TAG}_pmc_hbm_traffic.csv profiles/${TAG}_pmc_mfma_util.csv profiles/traffic.json profiles/mfma_util.json $O/
head -12 $O/${TAG}_kernel_stats_sde30.csv | cut -c1-160
