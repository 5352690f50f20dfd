#!/bin/bash
# usage: build_variant.sh NAME "-DFOO=1 ..."   -> tools/dbg/lib/libldt_NAME.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/dbg/lib /tmp/var_$1
for f in ldt_amd/csrc/*.hip; do
  extra=""; case "$(basename $f)" in fps_wave.hip) extra="-fno-slp-vectorize" ;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 $extra -c $f -o /tmp/var_$1/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib/libldt_$1.so /tmp/var_$1/*.o
