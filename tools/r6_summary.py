#!/usr/bin/env python3
"""Developer tool: the numbers README.md / DESIGN.md quote, from the evidence session's bench lines (gpurun_out/r6final or profiles/)."""
import json, sys, os
d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
def line(name):
    p = os.path.join(d, name)
    if not os.path.exists(p) or os.path.getsize(p) == 0: return None
    return json.loads(open(p).read().strip().splitlines()[-1])
for tag, name in (("C3", "r6_bench_line_driver_args.json"), ("C3 under rocprof", "r6_bench_line_under_rocprof.json"), ("C5", "r6_bench_line_c5.json"), ("C2", "r6_bench_line_c2.json")):
    j = line(name)
    if j is None: print(tag, "-"); continue
    r = j["roofline"]; k = r["kernels"]
    print(f"== {tag}: value {j['value']:.0f} {j['unit']} ms/step {j['ms_per_step']:.1f} steps {j['steps']} rerun {j.get('rerun_reads')} lanes {r.get('lanes_busy_of_64')} kernel_ms_of_step_ms {j.get('kernel_ms_of_step_ms')}")
    print(f"   kl_search frac {k['kl_search']['device_frac']} ms/launch {k['kl_search']['ms_per_launch']} x{k['kl_search']['launches']} Gvisits/s {k['kl_search']['Gvisits_per_s']} dev TB/step {k['kl_search']['device_bytes_per_step']/1e12:.2f}")
    if "kl_calc_d" in k: print(f"   kl_calc_d frac {k['kl_calc_d']['device_frac']} ms {k['kl_calc_d']['ms_per_launch']} dev TB/step {k['kl_calc_d']['device_bytes_per_step']/1e12:.2f}")
    print(f"   roofline frac {r['frac']} traffic {r.get('traffic')} traffic/dev {r.get('traffic_over_device_bytes')} s8d {json.dumps(r.get('s8d_check'))[:300]}")
    print(f"   traffic src {str(r.get('traffic_source'))[:400]}")
    for key in ("calculate_d_table", "cpu_baseline", "end_to_end", "also", "cli_end_to_end", "setup_s"):
        if key in j: print(f"   {key}: {json.dumps(j[key])[:1500]}")
