#!/bin/bash
# r2 probe 9: is the 43-of-64 lane occupancy at 884 M rows (400 M-char genome) the admission control of a too small pool?
set -u
mkdir -p gpurun_out/r2p9
show='
import sys, json
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernels"]
print("value", j["value"], "ms/step", j["ms_per_step"], "search ms/launch", k["kl_search"]["ms_per_launch"], "launches", k["kl_search"]["launches"], "frac", k["kl_search"]["frac"], "lanes", j["roofline"]["lanes_busy_of_64"], "rerun", j["rerun_reads"])'
for gb in "" 200; do
  echo "== BWB_POOL_GB=${gb:-default}"
  BWB_DEBUG=1 BWB_POOL_GB=$gb timeout 900 python bench.py --genome-mb 400 --pool 4000000 --reads 1000000 --steps 6 --warmup 1 --no-extras 2> gpurun_out/r2p9/dbg_${gb:-default}.log | grep '^{"metric"' | python3 -c "$show"
  grep "kl_search class 0" gpurun_out/r2p9/dbg_${gb:-default}.log | tail -3 | cut -c1-300
done 2>&1 | tee gpurun_out/r2p9/pool.log
