#!/bin/bash
# tools/dbg: the hand-scheduled 4-wave experiment (DMA instruction forms / cache policies) vs the shipped 256^2 kernel's main loop.
cd "$(dirname "$0")/../.."
B=tools/dbg/build
for rep in 1 2; do
  for v in $(ls $B | grep gemm4w_asm_ | sed 's/gemm4w_asm_//'); do
    for shape in "16384 4096 1024" "16384 1024 4096"; do
      timeout -k 5 60 $B/gemm4w_asm_$v $shape 0 || echo "FAILED $v $shape rc=$?"
    done
  done
done
DBG=1 timeout -k 5 120 python tools/dbg/gemm_bench.py 2>&1 | grep "up-discard\|dn-discard"
