#!/bin/bash
# tools/dbg: per-position kernel durations + gaps of the ViPC-conditioned loop (BASELINE configs[4]'s per-GPU share: B = 32, T = 32)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-x}
rm -rf gpurun_out/prof_c5seq_$TAG
rocprofv3 --kernel-trace -d gpurun_out/prof_c5seq_$TAG --output-format csv -- python3 tools/dbg/c5_prof.py > gpurun_out/prof_c5seq_$TAG.log 2>&1
f=$(find gpurun_out/prof_c5seq_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/dbg/trace_seq.py "$f" > gpurun_out/c5seq_$TAG.txt 2>&1
rm -rf gpurun_out/prof_c5seq_$TAG
cat gpurun_out/c5seq_$TAG.txt
