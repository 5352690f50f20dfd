#!/bin/bash
# usage: build_head_variant.sh [REV]   -> tools/dbg/lib/libldt_HEAD.so built from the committed sources of REV (default HEAD):
# the A/B partner of the working tree's library on the same box (LDT_HIP_LIB=tools/dbg/lib/libldt_HEAD.so).
set -e
cd "$(dirname "$0")/../.."
REV=${1:-HEAD}
rm -rf /tmp/head_src && mkdir -p /tmp/head_src/ldt_amd /tmp/head_src/include tools/dbg/lib
git archive "$REV" ldt_amd/csrc include | tar -x -C /tmp/head_src
for f in /tmp/head_src/ldt_amd/csrc/*.hip; do
  extra=""; case "$(basename $f)" in fps_wave.hip) extra="-fno-slp-vectorize" ;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c $f -o /tmp/head_src/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib/libldt_HEAD.so /tmp/head_src/*.o
echo built tools/dbg/lib/libldt_HEAD.so from $REV
