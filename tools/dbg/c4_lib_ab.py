#!/usr/bin/env python3
"""BASELINE configs[3] (tools/bench_compressor.py: encode / decode clouds/s + the two cross-attention microbench shapes) for two or more builds
of libldt_hip.so in alternating child processes on one box.   usage: c4_lib_ab.py libA.so libB.so ... [rounds]   ("product" = the in-tree build)"""
import json, os, subprocess, sys
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 2
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "product":
            env["LDT_HIP_LIB"] = l
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_compressor.py"), "--reps", "6"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            print(out.stdout[-500:], out.stderr[-1500:]); raise
        row = (d["encode_clouds_per_s"], d["decode_clouds_per_s"], d["cross_attn_q2048_kvT"]["us"], d["cross_attn_qT_kv2048"]["us"])
        res[l].append(row)
        print("round %d %-40s encode %8.1f  decode %8.1f clouds/s   q2048xkvT %6.1f us   qTxkv2048 %6.1f us" % ((r, os.path.basename(l)) + row), flush=True)
for l in libs:
    v = res[l]
    print("%-40s best: encode %8.1f  decode %8.1f   q2048xkvT %6.1f   qTxkv2048 %6.1f" % (os.path.basename(l), max(x[0] for x in v), max(x[1] for x in v), min(x[2] for x in v), min(x[3] for x in v)))
