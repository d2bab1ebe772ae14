"""Fold rocprofv3 --pmc passes (gpurun_out/<tag>_{fetch,write,sq1,sq2}) into profiles/<round>_pmc_counters.json and
profiles/traffic.json.  Per kernel only the dispatch with the largest grid (the full bench batch) is kept, so the
single-gate latency launches of bench.py do not blur the per-launch figures.

usage: python tools/summarise_pmc.py <tag> <round> [mode]     mode (exact | fast): the arithmetic the passes ran with;
it goes into the output file name and the traffic key that bench.py reads for that mode"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else None
names = {"k_blind_rotate": "blind_rotate", "k_keyswitch_mfma": "keyswitch_mfma", "k_ks_transpose": "ks_transpose"}
out = collections.defaultdict(dict)
for p in ("fetch", "write", "sq1", "sq2", "tcc"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, p), "*", "*_counter_collection.csv"))
    if not files:
        continue
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))   # gpurun_out/ keeps earlier runs of the tag
    for key, short in names.items():
        mine = [r for r in rows if key + "<" in r["Kernel_Name"] or key + "(" in r["Kernel_Name"]]
        if not mine:
            continue
        big = max(int(r["Grid_Size"]) for r in mine)
        disp = sorted({int(r["Dispatch_Id"]) for r in mine if int(r["Grid_Size"]) == big})[-1]
        for r in mine:
            if int(r["Dispatch_Id"]) == disp:
                out[short][r["Counter_Name"]] = out[short].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open(os.path.join(ROOT, "profiles", "%s_pmc_counters%s.json" % (rnd, "_" + mode if mode else "")), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
br = out["blind_rotate"]
if "FETCH_SIZE" in br and "WRITE_SIZE" in br:
    traffic = {
        "k_blind_rotate_hbm_bytes_per_launch": int((2 * br["FETCH_SIZE"] + br["WRITE_SIZE"]) * 1024),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, "
                  "B=16384, the full-batch dispatch only; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE "
                  "reports half of wide coalesced reads; MI355X_MICROARCH.md HBM section)",
        "FETCH_SIZE_KB": br["FETCH_SIZE"], "WRITE_SIZE_KB": br["WRITE_SIZE"],
        "note": "fabric-side (L2 miss) bytes; both keys (124 MB) stay resident in the 256 MB Infinity Cache, whose "
                "hits these counters include, so this is an upper bound on HBM bytes. Algorithmic bytes per launch: "
                "16384*7572 B of ciphertext I/O + 61.9 MB of key = 0.19 GB.",
    }
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if mode:   # keep the other mode's entry: one key per mode
        old = json.load(open(tpath)) if os.path.exists(tpath) else {}
        old["k_blind_rotate_%s_hbm_bytes_per_launch" % mode] = traffic["k_blind_rotate_hbm_bytes_per_launch"]
        old["%s_source" % mode] = traffic["source"] + " --arith " + mode + " (round " + rnd + ")"
        old["%s_FETCH_SIZE_KB" % mode], old["%s_WRITE_SIZE_KB" % mode] = br["FETCH_SIZE"], br["WRITE_SIZE"]
        traffic = old
    with open(tpath, "w") as f:
        json.dump(traffic, f, indent=1)
print(json.dumps(out, indent=1, sort_keys=True))
