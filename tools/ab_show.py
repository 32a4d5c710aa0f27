#!/usr/bin/env python3
"""Developer tool: one summary line of a bench.py JSON line (stdin), used by tools/ab_bench.sh."""
import json
import sys

name = sys.argv[1]
line = [l for l in sys.stdin.read().splitlines() if l.startswith('{"metric"')]
if not line:
    print(f"{name:14s} NO BENCH LINE")
    sys.exit(0)
j = json.loads(line[-1])
r = j["roofline"]
k = r["kernels"]
s, d = k["kl_search"], k["kl_calc_d"]
print(f"{name:14s} value {j['value']:10.1f} ms/step {j['ms_per_step']:9.1f} | search ms/launch {s['ms_per_launch']:9.1f} x{s['launches']} "
      f"dev_frac {s['device_frac']:.4f} {s.get('Gvisits_per_s', 0):6.2f} Gvisits/s dev_bytes/step {s['device_bytes_per_step'] / 1e12:.2f} TB"
      f" | calc_d ms {d['ms_per_launch']:8.1f} dev_frac {d['device_frac']:.4f} | lanes {r['lanes_busy_of_64']} rerun {j['rerun_reads']}")
