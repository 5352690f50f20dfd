#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE) to
profiles/<tag>_pmc_hbm_traffic.csv and profiles/traffic.json (HBM bytes per launch per kernel).

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled;
WRITE_SIZE exact for 16-B streaming stores; both in KiB.   usage: reduce_pmc.py TAG FETCH_DIR WRITE_DIR NOTE"""
import collections, csv, glob, json, os, re, sys

tag, fdir, wdir, note = sys.argv[1:5]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha  # noqa: E402  (sha of ldt_amd/csrc at collection time: bench.py withholds a stale measurement)
SHA = csrc_sha()


def load(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
                name = re.sub(r"\(.*$", "", name)
                out[name].append(float(r["Counter_Value"]))
    return out


fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
rows, traffic = [], {}
for counter, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
    for k, v in sorted(d.items()):
        rows.append((counter, k, len(v), sum(v) / len(v), min(v), max(v)))
for k in sorted(set(fetch) & set(write)):
    f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
    traffic[k] = {"fetch_size_kib_raw": round(f, 1), "write_size_kib": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024), "note": note,
                  "source": "profiles/%s_pmc_hbm_traffic.csv" % tag, "csrc_sha": SHA}
with open(os.path.join(ROOT, "profiles", tag + "_pmc_hbm_traffic.csv"), "w") as fo:
    fo.write("counter,kernel,launches,avg_KiB,min_KiB,max_KiB\n")
    for r in rows:
        fo.write("%s,\"%s\",%d,%.1f,%.1f,%.1f\n" % r)
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fo:
    json.dump(traffic, fo, indent=1, sort_keys=True)
print("kernels:", len(traffic))
