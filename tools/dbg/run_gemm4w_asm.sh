#!/bin/bash
# tools/dbg: the hand-scheduled 4-wave experiment vs the shipped 256^2 kernel's main loop, one box, alternating.
cd "$(dirname "$0")/../.."
B=tools/dbg/build
for rep in 1 2; do
  for v in s1m1 s0m1 s1m0; do
    for shape in "16384 4096 1024" "16384 1024 4096" "16384 3072 1024" "16384 1024 1024"; do
      timeout -k 5 60 $B/gemm4w_asm_$v $shape 0 || echo "FAILED $v $shape rc=$?"
    done
  done
  DBG=1 timeout -k 5 120 python tools/dbg/gemm_bench.py
done
for v in s1m1; do
  for shape in "16384 4096 1024" "16384 1024 4096"; do timeout -k 5 60 $B/gemm4w_asm_$v $shape 1; done
done
