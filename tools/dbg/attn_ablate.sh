#!/bin/bash
# tools/dbg: whole-head attention kernel with the loads / the compute compiled out (variants built by build_variant.sh)
cd "$(dirname "$0")/../.."
for lib in product tools/dbg/lib/libldt_att_noload.so tools/dbg/lib/libldt_att_nocomp.so; do
  echo "== $lib"
  if [ "$lib" = product ]; then LDT_ATTN_FORCE=3 python3 tools/dbg/attn_head_ab.py child 2>/dev/null
  else LDT_HIP_LIB=$lib LDT_ATTN_FORCE=3 python3 tools/dbg/attn_head_ab.py child 2>/dev/null; fi
done
