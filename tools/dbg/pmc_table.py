#!/usr/bin/env python3
"""tools/dbg: per-kernel averages of every counter in a rocprofv3 --pmc output directory (one or more passes).
usage: pmc_table.py DIR [DIR ...]   (prints a table; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024) where both are present)"""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for d in acc.values() for c in d})
print("%-58s %6s " % ("kernel", "calls") + " ".join("%24s" % c for c in counters) + "  derived")
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if k.startswith("at::") or k.startswith("__amd") or not k: continue
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    n = max(len(v) for v in d.values())
    extra = []
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("GRBM_GUI_ACTIVE"):
        extra.append("mfma_util %.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
    if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
        extra.append("lds_conflict/active %.3f" % (avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]))
    if avg.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_ANY" in avg:          # disjoint shares of a wave's cycles (MI355X_MICROARCH.md, SQ counters)
        wc = avg["SQ_WAVE_CYCLES"]
        extra.append("wave cycles: issuing %.3f issue-stalled %.3f parked %.3f" % (avg["SQ_ACTIVE_INST_ANY"] / wc, avg.get("SQ_WAIT_INST_ANY", float("nan")) / wc, avg.get("SQ_WAIT_ANY", float("nan")) / wc))
        if avg.get("SQ_WAVES"): extra.append("kcycles/wave %.1f" % (4e-3 * wc / avg["SQ_WAVES"]))
    if "FETCH_SIZE" in avg: extra.append("fetch %.1f MB" % (avg["FETCH_SIZE"] * 1024 / 1e6))
    if "WRITE_SIZE" in avg: extra.append("write %.1f MB" % (avg["WRITE_SIZE"] * 1024 / 1e6))
    print("%-58s %6d " % (k[:58], n) + " ".join("%24.0f" % avg.get(c, float("nan")) for c in counters) + "  " + "; ".join(extra))
