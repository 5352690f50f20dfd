#!/bin/bash
# tools/dbg: per-position kernel durations + gaps of the T = 256 bench loop (eager launches from the C++ loop: are there gaps?)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}
rm -rf gpurun_out/prof_t256seq_$TAG
rocprofv3 --kernel-trace -d gpurun_out/prof_t256seq_$TAG --output-format csv -- python3 bench.py --sde-steps 30 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-roofline > gpurun_out/prof_t256seq_$TAG.log 2>&1
f=$(find gpurun_out/prof_t256seq_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/dbg/trace_seq.py "$f" > gpurun_out/t256seq_$TAG.txt 2>&1
rm -rf gpurun_out/prof_t256seq_$TAG
head -14 gpurun_out/t256seq_$TAG.txt; tail -12 gpurun_out/t256seq_$TAG.txt
