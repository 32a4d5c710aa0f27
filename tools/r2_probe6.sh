#!/bin/bash
# r2 probe 6: A/B at C3 (workload built once): next-top entry fetched at pop time (product) against at the end of the iteration.
set -u
mkdir -p gpurun_out/r2p6
show='
import sys, json
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernels"]
print("value", j["value"], "ms/step", j["ms_per_step"], "search ms/launch", k["kl_search"]["ms_per_launch"], "launches", k["kl_search"]["launches"], "frac", k["kl_search"]["frac"], "lanes", j["roofline"]["lanes_busy_of_64"], "calc_d ms", k["kl_calc_d"]["ms_per_launch"], "rerun", j["rerun_reads"])'
for lib in "" bwbble_amd/tools_exp/libbwbble_hip_noearlytop.so ""; do
  echo "== lib ${lib:-product}"
  BWB_LIB=${lib:+$PWD/$lib} timeout 1500 python bench.py --steps 6 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | python3 -c "$show"
done 2>&1 | tee gpurun_out/r2p6/ab.log
