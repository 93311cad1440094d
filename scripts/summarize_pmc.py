#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (one directory per pass) -> one JSON with the
per-launch averages of the engine's kernels.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in
KiB; they are converted to bytes here.  On gfx950 FETCH_SIZE reports half the bytes of wide (16 B per lane) reads:
kernels whose row reads are 16 B per lane (sgd_wide_kernel, score_screen_kernel on a table that fills the query width)
get fetch_bytes_corrected = 2 x FETCH_SIZE and their traffic is computed from it; for the others (4 or 8 B per lane,
an access width the counter is not calibrated for) the value is kept as measured and flagged."""
import collections
import csv
import glob
import json
import sys

out_path, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if any(k in name for k in ("sgd_", "sampler_kernel", "score_count", "score_screen", "screen16_", "correct_clicked", "build_queries", "part_")):
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in agg.items():
    e = {c: sum(x) / len(x) for c, x in v.items()}
    e["launches_seen"] = max(len(x) for x in v.values())
    if "FETCH_SIZE" in e:
        e["fetch_bytes"] = e["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024
    wide = "sgd_wide_kernel" in k or "score_screen_kernel" in k or "score_screen16_kernel" in k
    if "fetch_bytes" in e and wide:
        e["fetch_bytes_corrected"] = 2 * e["fetch_bytes"]
    if "fetch_bytes" in e and "write_bytes" in e:
        e["hbm_traffic_bytes_per_launch"] = e.get("fetch_bytes_corrected", e["fetch_bytes"]) + e["write_bytes"]
        e["note"] = ("16 B per lane reads: FETCH_SIZE doubled (gfx950 tallies their 128-B requests at 64 B); WRITE_SIZE includes float atomics" if wide else
                     "FETCH_SIZE as measured (4 / 8 B per lane loads: uncalibrated width; x2 would be the upper bound); WRITE_SIZE includes float atomics")
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    res[k] = e
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: {c: round(x, 1) for c, x in v.items() if isinstance(x, float)} for k, v in res.items()}, indent=1)[:3000])
