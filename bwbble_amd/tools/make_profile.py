#!/usr/bin/env python3
"""Assembles profiles/<tag>_bench_profile.json, <tag>_bench_kernel_stats.csv and <tag>_bench_line.json from the rocprofv3
output directories written by make_profile.sh.  usage: make_profile.py <rocprof_out_dir> <profiles_dir> <tag>"""
import csv, glob, json, os, shutil, sys
out, prof, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def short(name):
    return name.split("(")[0].replace("void ", "")


res = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 ; "
                  "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 (separate passes)",
       "workload": "C2 chr21-scale synthetic multi-genome, 1 M x 100 bp reads, align -n 3, then the same batch at -n 0"}
stats = one("trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
trace = one("trace/**/*kernel_trace.csv")
if trace:
    rows = []
    for r in csv.DictReader(open(trace)):
        if "kl_" in r["Kernel_Name"]:
            rows.append({"kernel": short(r["Kernel_Name"]), "dispatch": int(r["Dispatch_Id"]), "grid": int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]),
                         "vgpr": int(r.get("VGPR_Count", 0) or 0), "accum_vgpr": int(r.get("Accum_VGPR_Count", 0) or 0), "sgpr": int(r.get("SGPR_Count", 0) or 0),
                         "lds": int(r.get("LDS_Block_Size", 0) or 0), "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    res["kernel_trace_launches"] = rows
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = one(f"pmc_{c}/**/*counter_collection.csv")
    if not f:
        continue
    agg = {}
    for r in csv.DictReader(open(f)):
        if "kl_" in r["Kernel_Name"] and r["Counter_Name"] == c:
            k = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    res[f"pmc_{c}_KB_per_launch"] = [{"kernel": k[1], "dispatch": k[0], "value": v} for k, v in sorted(agg.items())]


def first(counter, kernel):  # the first launch of `kernel` in the pass: the -n 3 batch
    for e in res.get(f"pmc_{counter}_KB_per_launch", []):
        if e["kernel"].startswith(kernel):
            return e["value"]
    return None


note = ("gfx950: FETCH_SIZE counts 128-B fabric reads at 64 B (MI355X_MICROARCH.md, HBM section); calibrated on this access pattern "
        "(one lane = 8 x dwordx4 of one 128-B bucket, 1 GiB table, bwbble_amd/tools_exp/lane_bench, "
        "profiles/r1_fetch_size_calibration_lane_bench.csv): FETCH_SIZE*1024 / known bytes = 0.50, so reads are doubled; WRITE_SIZE is "
        "taken as is. Infinity-Cache hits are included (the 106 MB index is cache resident).")
for key, kern in (("kl_search_n3_launch", "kl_search"), ("kl_calc_d_launch", "kl_calc_d")):
    f, w = first("FETCH_SIZE", kern), first("WRITE_SIZE", kern)
    if f is not None and w is not None:
        res[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_read_bytes_corrected": f * 1024 * 2, "hbm_write_bytes": w * 1024, "note": note}
json.dump(res, open(os.path.join(prof, f"{tag}_bench_profile.json"), "w"), indent=1)
line = os.path.join(out, "bench_line.json")
if os.path.exists(line):
    txt = [l for l in open(line) if l.startswith("{")]
    if txt:
        open(os.path.join(prof, f"{tag}_bench_line.json"), "w").write(txt[-1])
print("wrote", os.path.join(prof, f"{tag}_bench_profile.json"))
