#!/bin/bash
# tools/dbg: per-position kernel durations + gaps of the T = 32 sampling loop
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}
rm -rf gpurun_out/prof_t32seq_$TAG
rocprofv3 --kernel-trace -d gpurun_out/prof_t32seq_$TAG --output-format csv -- python3 tools/dbg/t32_prof.py > gpurun_out/prof_t32seq_$TAG.log 2>&1
f=$(find gpurun_out/prof_t32seq_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/dbg/trace_seq.py "$f" > gpurun_out/t32seq_$TAG.txt 2>&1
rm -rf gpurun_out/prof_t32seq_$TAG
cat gpurun_out/t32seq_$TAG.txt
