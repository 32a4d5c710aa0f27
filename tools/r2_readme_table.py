#!/usr/bin/env python3
"""Prints the 'Measured' table of README.md from the committed bench lines in profiles/ (so the numbers are not hand-typed)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    p = os.path.join(ROOT, "profiles", name)
    return json.loads(open(p).read().strip().splitlines()[-1]) if os.path.exists(p) and os.path.getsize(p) else None


rows = []
for name, label in (("r2_bench_line_driver_args.json", "`bench.py --gpus 1 --steps 20 --warmup 5` (the driver's arguments)"), ("r2_bench_line.json", "`bench.py` (3 steps + 1 warm-up: one drain in four launches)")):
    j = line(name)
    if not j:
        continue
    k = j["roofline"]["kernels"]
    cpu = j.get("cpu_baseline", {})
    rows.append(f"| {label} | {j['value'] / 1e3:.0f} k | {j['also']['n0']['value'] / 1e6:.2f} M | `kl_search` {k['kl_search']['frac']:.2f} (device {k['kl_search']['device_frac']:.2f}), "
                f"`kl_calc_d` {k['kl_calc_d']['frac']:.2f} (device {k['kl_calc_d']['device_frac']:.2f}) | {j['roofline']['lanes_busy_of_64']} | "
                f"{j['end_to_end']['of_value']:.2f} | {cpu.get('value', 0):.0f} reads/s on {cpu.get('cores')} threads ({j['value'] / max(cpu.get('value', 1), 1):.0f}x), sample parity {cpu.get('parity_on_sample')} |")
print("| GRCh37-scale run (6.85 G rows, u64, 2.5 M-read steps) | `-n 3` reads/s | `-n 0` reads/s | fraction of 8 TB/s: algorithmic 192 B/visit (device bytes) | lanes busy of 64 | end-to-end / value | reference on the same box |")
print("|---|---|---|---|---|---|---|")
print("\n".join(rows))
j = line("r2_bench_line.json")
if j:
    m = j["rank_micro"]
    print(f"\nRank micro-benchmark on the same index (random Occ16, {m['queries']} queries): octet layout {m['octet_layout']['Gvisits_per_s']} G visits/s "
          f"(device {m['octet_layout']['device_frac']:.2f} of 8 TB/s), cooperative gather + per-lane rank {m['lane_layout']['Gvisits_per_s']} G visits/s "
          f"(device **{m['lane_layout']['device_frac']:.2f}**).")
