#!/usr/bin/env python3
"""Developer tool: every launch of the alignment kernels from a rocprofv3 --kernel-trace CSV, next to the bench line the same run printed.
usage: kernel_launches.py <run_kernel_trace.csv> <bench_line.json> <out.json> [description]"""
import csv, json, sys
trace, line, out = sys.argv[1:4]
what = sys.argv[4] if len(sys.argv) > 4 else ""
k = {}
for r in csv.DictReader(open(trace)):
    name = r.get("Kernel_Name", "")
    for key in ("kl_calc_d", "kl_search"):
        if key in name:
            d = k.setdefault(key, {"name": name.split("(")[0], "grid": int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), "workgroup": int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), "t": []})
            d["t"].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
res = {"what": what, "kernels": {}}
for key, d in k.items():
    ms = [round(v, 3) for _, v in sorted(d["t"])]
    res["kernels"][key] = {"name": d["name"], "grid": d["grid"], "workgroup": d["workgroup"], "ms": ms, "launches": len(ms), "avg_ms": round(sum(ms) / max(len(ms), 1), 3)}
try:
    j = json.loads([ln for ln in open(line) if ln.startswith('{"metric"')][-1])
    ks = j["roofline"]["kernels"]
    res["bench_line_live_hip_events"] = {n: {"launches": ks[n]["launches"], "ms_per_launch": ks[n]["ms_per_launch"]} for n in ks}
    # the timed region = the LAST `launches` launches of each kernel in the trace (the warm-up's come first)
    res["timed_region_from_trace"] = {}
    for n in ks:
        ms = res["kernels"].get(n, {}).get("ms", [])
        m = ks[n]["launches"]
        if ms and m:
            res["timed_region_from_trace"][n + "_ms_per_launch"] = round(sum(ms[-m:]) / m, 3)
    res["launch_order"] = "warm-up step(s) (slices + the draining launch at their flush), then the timed region = its slices + the draining launch of the final flush; kl_calc_d: one launch per step"
except Exception as e:  # noqa: BLE001
    res["bench_line_error"] = str(e)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res.get("timed_region_from_trace", {})), json.dumps(res.get("bench_line_live_hip_events", {})))
