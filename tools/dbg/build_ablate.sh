#!/bin/bash
# timing-only debug builds of the GEMM with pieces of the main loop removed (results are WRONG by design)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/dbg/lib
for a in ${ABL:-1 2 3 4 7}; do
  mkdir -p /tmp/abl$a
  for f in ldt_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DV2_ABLATE=$a -c $f -o /tmp/abl$a/$(basename ${f%.hip}).o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib/libldt_abl$a.so /tmp/abl$a/*.o
done
ls -la tools/dbg/lib
