#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/r2_profile.sh into the files committed under profiles/ (r2_*)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    for k in ("kl_search", "kl_calc_d", "k_dseed_inherit", "k_relayout", "k_rank_bench_lane", "k_rank_bench"):
        if k in name:
            return k
    return name[:40]


# per-kernel stats and the per-launch list from the kernel trace
for f in glob.glob(src + "/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(dst, "r2_c3_kernel_stats.csv"))
launches = collections.defaultdict(list)
for f in glob.glob(src + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        launches[short(r["Kernel_Name"])].append(round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
line = None
for name in ("r2_bench_line.json", "r2_bench_line_driver_args.json", "r2_bench_line_under_rocprof.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, name))
        if name == "r2_bench_line.json":
            line = json.loads(open(p).read().strip().splitlines()[-1])

# PMC passes: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies a 128-byte fabric read as 64 bytes (MI355X_MICROARCH.md, HBM section;
# calibrated on this kernel's access shape in round 1, profiles/README.md): reads x2
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(src + f"/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    pmc[c] = {k: (v, len(n[k])) for k, v in agg.items()}
out = {"what": "rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 0 --no-extras` (default workload: C3), one counter per pass",
       "genome_mb": 3100.0, "reads": 2500000, "ndiff": 3, "steps_in_pass": 2,
       "correction": "hbm bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 counts a 128-B fabric read as 64 B) + WRITE_SIZE KiB x 1024"}
for k in ("kl_search", "kl_calc_d"):
    if k in pmc.get("FETCH_SIZE", {}) and k in pmc.get("WRITE_SIZE", {}):
        rd, nl = pmc["FETCH_SIZE"][k]
        wr, _ = pmc["WRITE_SIZE"][k]
        tot = rd * 1024 * 2 + wr * 1024
        out[k] = {"launches_in_pass": nl, "hbm_read_bytes_corrected": rd * 1024 * 2, "hbm_write_bytes": wr * 1024,
                  "hbm_bytes_per_step": tot / 2, "hbm_bytes_per_launch": tot / max(nl, 1)}
if line:
    for k in ("kl_search", "kl_calc_d"):
        if k in out:
            kk = line["roofline"]["kernels"][k]
            alg = kk["visits_per_step"] * 192
            out[k]["algorithmic_bytes_per_step"] = alg
            out[k]["device_bytes_per_step"] = kk["device_bytes_per_step"]
            out[k]["traffic_over_algorithmic"] = round(out[k]["hbm_bytes_per_step"] / alg, 3)
            out[k]["traffic_over_device_bytes"] = round(out[k]["hbm_bytes_per_step"] / kk["device_bytes_per_step"], 3)
            if "bucket_bytes_per_step" in kk:  # x2 only for the 128-byte bucket requests, the 64-byte metadata requests as counted
                raw_read = out[k]["hbm_read_bytes_corrected"] / 2 / 2
                out[k]["hbm_bytes_per_step_buckets_only_doubled"] = raw_read + kk["bucket_bytes_per_step"] / 2 + out[k]["hbm_write_bytes"] / 2
json.dump(out, open(os.path.join(dst, "r2_c3_pmc.json"), "w"), indent=1)
json.dump({"per_launch_ms": launches, "source": "rocprofv3 --kernel-trace of bench.py --steps 3 --warmup 1 --no-extras (C3)"},
          open(os.path.join(dst, "r2_c3_kernel_launches.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
