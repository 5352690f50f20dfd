#!/bin/bash
# Builds libldt_hip.so (all HIP kernels + the C-ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
# Links EXACTLY the objects of the *.hip files present (an object whose source is gone is deleted, never linked), then runs the
# ISA lint (isa_lint.py: hand-counted waits / hazard padding of the asm-scheduled kernels) over the gfx950 disassembly.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libldt_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p "$HERE/build"
for o in "$HERE"/build/*.o; do                        # orphans: objects without a source (a removed experiment must not ship)
  [ -e "$o" ] || continue
  [ -f "$HERE/$(basename "${o%.o}").hip" ] || { echo "removing orphan object $(basename "$o")"; rm -f "$o"; }
done
pids=()
objs=()
for f in "$HERE"/*.hip; do
  o="$HERE/build/$(basename "${f%.hip}").o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/kernels.h" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/attn_tile.h" -nt "$o" ] || [ "$HERE/../../include/ldt_hip.h" -nt "$o" ]; then
    extra=""
    case "$(basename "$f")" in fps_wave.hip) extra="-fno-slp-vectorize" ;; esac   # (why: the file's header)
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"      # always relinked (~1 s): the .so is a function of the sources present
if [ "${LDT_SKIP_ISA_LINT:-0}" != 1 ]; then
  python3 "$HERE/isa_lint.py" "$OUT"
fi
echo "built $OUT"
