#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE) to
profiles/<tag>_pmc_mfma_util.csv and profiles/mfma_util.json (per kernel symbol, with the kernel-source sha it was collected on).

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): GRBM_GUI_ACTIVE is summed over the 8 XCDs (so / 8 = the
dispatch's cycles), the chip has 1024 SIMDs (MI355X_MICROARCH.md, rocprofv3 PMC slots / DVFS give-back).  It is the fraction of
SIMD-cycles with the matrix pipe busy AT THE CLOCK THE CHIP HELD during the dispatch (profiled passes run at 1.9-1.95 GHz).
usage: reduce_pmc_mfma.py TAG PMC_DIR NOTE"""
import collections, csv, glob, json, os, re, sys

tag, pdir, note = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha  # noqa: E402

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", ""))
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows, out = [], {}
for k, d in sorted(acc.items()):
    need = ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    if not all(n in d for n in need):
        continue
    avg = {n: sum(v) / len(v) for n, v in d.items()}
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    util = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0) if cyc > 0 else 0.0
    wait = avg.get("SQ_WAIT_ANY", 0.0) * 4.0 / (avg.get("SQ_BUSY_CYCLES", 0.0) or 1.0)
    rows.append((k, len(d["GRBM_GUI_ACTIVE"]), avg["SQ_VALU_MFMA_BUSY_CYCLES"], avg.get("SQ_BUSY_CYCLES", 0.0), avg.get("SQ_WAIT_ANY", 0.0), avg["GRBM_GUI_ACTIVE"], util))
    out[k] = {"mfma_util": round(util, 4), "launches": len(d["GRBM_GUI_ACTIVE"]), "dispatch_cycles": int(cyc), "note": note,
              "source": "profiles/%s_pmc_mfma_util.csv" % tag, "csrc_sha": csrc_sha()}
with open(os.path.join(ROOT, "profiles", tag + "_pmc_mfma_util.csv"), "w") as fo:
    fo.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,GRBM_GUI_ACTIVE,mfma_util\n")
    for r in rows:
        fo.write("\"%s\",%d,%.0f,%.0f,%.0f,%.0f,%.4f\n" % r)
with open(os.path.join(ROOT, "profiles", "mfma_util.json"), "w") as fo:
    json.dump(out, fo, indent=1, sort_keys=True)
for r in sorted(rows, key=lambda r: -r[2])[:8]:
    print("%-60s launches %4d  mfma_util %.3f" % (r[0][:60], r[1], r[6]))
