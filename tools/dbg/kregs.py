#!/usr/bin/env python3
"""Register / LDS / scratch usage of the gfx950 kernels in a built library.   usage: kregs.py [lib.so] [regex]"""
import os, re, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "ldt_amd", "csrc"))
import isa_lint as L
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "..", "ldt_amd", "libldt_hip.so")
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for elf in L.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf); f.flush()
        meta = L.kernel_meta(f.name)
    names = L.demangle(list(meta))
    for k, m in sorted(meta.items(), key=lambda kv: names[kv[0]]):
        n = names[k]
        if pat.search(n):
            print("%-90s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %d spill %d" % (n[:90], m.get(".vgpr_count", 0), m.get(".agpr_count", 0),
                  m.get(".sgpr_count", 0), m.get(".group_segment_fixed_size", 0), m.get(".private_segment_fixed_size", 0), m.get(".vgpr_spill_count", 0)))
