#!/bin/bash
# Builds libldt_hip.so (all HIP kernels + the C-ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libldt_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/build"
pids=()
for f in "$HERE"/*.hip; do
  o="$HERE/build/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/kernels.h" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/attn_tile.h" -nt "$o" ] || [ "$HERE/../../include/ldt_hip.h" -nt "$o" ]; then
    extra=""
    case "$(basename "$f")" in fps_wave.hip) extra="-fno-slp-vectorize" ;; esac   # (why: the file's header)
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE"/build/*.o
echo "built $OUT"
