#!/bin/bash
# tools/dbg: the full-line-DMA 4-wave experiment and its ablations vs the shipped 256^2 kernel's main loop, one box.
cd "$(dirname "$0")/../.."
B=tools/dbg/build
for rep in 1 2; do
  for v in base nomfma noread nodma; do
    for shape in "16384 4096 1024" "16384 1024 4096"; do
      timeout -k 5 60 $B/gemm4w_fl_$v $shape 0 || echo "FAILED $v $shape rc=$?"
    done
  done
done
for shape in "16384 3072 1024" "16384 1024 1024" "16384 1024 8192" "8192 1024 4096" "2048 4096 1024"; do timeout -k 5 60 $B/gemm4w_fl_base $shape 0; done
for shape in "16384 4096 1024" "16384 1024 4096" "16384 3072 1024"; do timeout -k 5 60 $B/gemm4w_fl_base $shape 1; done
DBG=1 timeout -k 5 120 python tools/dbg/gemm_bench.py 2>&1 | grep "discard\|up-bf16" | cut -c1-80
