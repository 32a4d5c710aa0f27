#!/usr/bin/env python3
"""Developer tool: turns the CSVs of tools/pmc_traffic.sh into profiles/<tag>_pmc.json.

Model of the read side (MI355X_MICROARCH.md: FETCH_SIZE derives from the L2's memory-side request counter and tallies a 128-byte
request as 64 bytes on gfx950): every request counted by TCC_EA0_RDREQ moves 64 bytes (32 for the ones also counted by _32B),
except the 128-byte bucket requests, which move 128.  The calibration pass measures exactly that on known request counts (requests
per bucket gather, per 8-byte and per 16-byte scattered load); the number of bucket requests of a bench launch is counted in the
kernel (bucket_loads_*).  hbm_read = (RDREQ - RDREQ_32B) x 64 + RDREQ_32B x 32 + buckets x 64 x [buckets cost one RDREQ each]."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

out_dir, dst = sys.argv[1], sys.argv[2]
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "kl_search" if "kl_search" in k else "kl_calc_d" if "kl_calc_d" in k else "k_coop" if "k_coop" in k else \
                   "k_meta8" if "k_meta" in k and "8" in k.split("k_meta")[1][:6] and "16" not in k.split("k_meta")[1][:6] else "k_meta16" if "k_meta" in k else None
            if name is None:
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[name].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return agg, {k: len(v) for k, v in launches.items()}


def per_dispatch(sub):
    """{kernel: [counters of every dispatch, in dispatch order]}"""
    rows = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "kl_search" if "kl_search" in k else "kl_calc_d" if "kl_calc_d" in k else None
            if name is None:
                continue
            rows[name][int(r.get("Dispatch_Id", 0) or 0)][r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: [v[d] for d in sorted(v)] for k, v in rows.items()}


def launch_log(name):
    p = os.path.join(out_dir, name)
    return [json.loads(l) for l in open(p)] if os.path.exists(p) else []


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bwbble_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


cal, _ = counters("calib")
N = float(1 << 27)
calib = {}
for k in ("k_coop", "k_meta8", "k_meta16"):
    c = cal.get(k, {})
    if c:
        calib[k] = {"requests_issued": N, "RDREQ_per_request": round(c.get("TCC_EA0_RDREQ_sum", 0) / N, 4), "RDREQ_32B_per_request": round(c.get("TCC_EA0_RDREQ_32B_sum", 0) / N, 4),
                    "BUBBLE_per_request": round(c.get("TCC_BUBBLE_sum", 0) / N, 4), "RDREQ_DRAM_per_request": round(c.get("TCC_EA0_RDREQ_DRAM_sum", 0) / N, 4)}
rd, nl = counters("rd")
wr, _ = counters("wr")
line = [l for l in open(os.path.join(out_dir, "rd.json")).read().splitlines() if l.startswith('{"metric"')]
bench = json.loads(line[-1]) if line else {}
cfg = bench.get("config", {})
res = {"what": f"rocprofv3 --pmc passes of `bench.py --steps {STEPS} --warmup 0 --no-extras` (tools/pmc_traffic.sh), read-side raw counters in one pass, write side in another",
       "source_hash": source_hash(), "genome_mb": cfg.get("genome_mb"), "reads": cfg.get("reads_per_gpu_per_step"),
       "ndiff": cfg.get("max_diff"), "read_len": cfg.get("read_len"), "steps_in_pass": STEPS, "calibration_7GiB_table": calib,
       "method": "hbm read bytes = (RDREQ - RDREQ_32B) x 64 + RDREQ_32B x 32 + in-kernel bucket count x 64 (a 128-byte bucket request is one RDREQ, tallied at 64 bytes: "
                 "see calibration_7GiB_table.k_coop; a scattered 8- or 16-byte load is one 64-byte request: k_meta8/16), + WRITE_SIZE"}
kern = bench.get("roofline", {}).get("kernels", {})
for k in ("kl_search", "kl_calc_d"):
    if k not in rd:
        continue
    c, w = rd[k], wr.get(k, {})
    launches = max(nl.get(k, 1), 1)
    steps = STEPS
    bkt_bytes = kern.get(k, {}).get("bucket_bytes_per_step", 0) * steps
    raw = (c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    read = raw + bkt_bytes / 2  # the second 64 bytes of every 128-byte bucket request
    write = w.get("WRITE_SIZE", 0) * 1024
    dev = kern.get(k, {}).get("device_bytes_per_step", 0) * steps
    res[k] = {"launches_in_pass": launches, "RDREQ": c.get("TCC_EA0_RDREQ_sum"), "RDREQ_32B": c.get("TCC_EA0_RDREQ_32B_sum"), "BUBBLE": c.get("TCC_BUBBLE_sum"),
              "RDREQ_DRAM": c.get("TCC_EA0_RDREQ_DRAM_sum"), "WRREQ": w.get("TCC_EA0_WRREQ_sum"), "WRREQ_64B": w.get("TCC_EA0_WRREQ_64B_sum"),
              "read_bytes_as_counted": raw, "bucket_bytes_in_kernel": bkt_bytes, "hbm_read_bytes": read, "hbm_write_bytes": write,
              "hbm_bytes_per_step": (read + write) / steps, "hbm_bytes_per_launch": (read + write) / launches,
              "device_bytes_per_step": dev / steps, "traffic_over_device_bytes": round((read + write) / dev, 3) if dev else None,
              "kernel_ms_per_launch_in_pass": kern.get(k, {}).get("ms_per_launch")}
# (round 6) every dispatch priced on its own: the read pass's dispatches of a kernel, in order, are the launches of its launch log (the library
# wrote one line per launch: buckets, entries, records, whether the launch drains); the write pass runs the same launches in the same order
rd_d, wr_d = per_dispatch("rd"), per_dispatch("wr")
rd_log, wr_log = launch_log("rd_launches.jsonl"), launch_log("wr_launches.jsonl")
_fl = cfg.get("flags", "").split()
_opt = lambda name, dflt: int(_fl[_fl.index(name) + 1]) if name in _fl else dflt
esz = 32 if (_opt("-o", 1) > 1 or max(_opt("-M", 3), _opt("-O", 11), _opt("-E", 4)) > 63) else 16  # (bwb_hip.hip slot_upload: `wide`)
for k in ("kl_search", "kl_calc_d"):
    logs = [l for l in rd_log if l["kernel"] == k and l["class"] == 0]
    rows, wrows = rd_d.get(k, []), wr_d.get(k, [])
    if k not in res or not logs or len(logs) != len(rows) or len(wrows) != len(rows):
        if k in res:
            res[k]["per_launch"] = f"not available: {len(logs)} log lines, {len(rows)} read-pass dispatches, {len(wrows)} write-pass dispatches"
        continue
    per = []
    for l, c, w in zip(logs, rows, wrows):
        raw = (c.get("TCC_EA0_RDREQ_sum", 0) - c.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
        read = raw + l["buckets"] * 64
        write = w.get("WRITE_SIZE", 0) * 1024
        dev = l["buckets"] * 128 + (l["entries_stored"] + l["entries_loaded"]) * esz + l["records_loaded"] * 16
        per.append({"drains": l["drains"], "ms": l["ms"], "buckets": l["buckets"], "RDREQ": c.get("TCC_EA0_RDREQ_sum"), "WRREQ": w.get("TCC_EA0_WRREQ_sum"),
                    "hbm_read_bytes": read, "hbm_write_bytes": write, "hbm_bytes": read + write, "device_bytes": dev})
    sl, dr = [q for q in per if not q["drains"]], [q for q in per if q["drains"]]
    mean = lambda v, f: sum(q[f] for q in v) / len(v) if v else None
    res[k]["per_launch"] = {"launches": per,
                            "slice": {"launches": len(sl), "hbm_bytes_per_launch": mean(sl, "hbm_bytes"), "device_bytes_per_launch": mean(sl, "device_bytes"), "ms_per_launch": mean(sl, "ms"),
                                      "RDREQ_minus_buckets_per_launch": (mean(sl, "RDREQ") - mean(sl, "buckets")) if sl else None},
                            "drain": {"launches": len(dr), "hbm_bytes_per_launch": mean(dr, "hbm_bytes"), "device_bytes_per_launch": mean(dr, "device_bytes"), "ms_per_launch": mean(dr, "ms")}}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != 'per_launch'} if isinstance(v, dict) else v) for k, v in res.items()}, indent=1)[:3000])
for k in ('kl_search', 'kl_calc_d'):
    pl = res.get(k, {}).get('per_launch')
    if isinstance(pl, dict):
        print(k, 'slice', pl['slice'], 'drain', pl['drain'])
    elif pl:
        print(k, pl)
