#!/bin/bash
# r2 probe 13: A/B at C3: speculative gather of the next pop's bucket (tools_exp/libbwbble_hip_spec.so, -DBWB_SPEC_PREFETCH) against the product.
set -u
mkdir -p gpurun_out/r2p13
show='
import sys, json
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernels"]
print("value", j["value"], "ms/step", j["ms_per_step"], "search ms/launch", k["kl_search"]["ms_per_launch"], "launches", k["kl_search"]["launches"], "frac", k["kl_search"]["frac"], "lanes", j["roofline"]["lanes_busy_of_64"], "calc_d ms", k["kl_calc_d"]["ms_per_launch"], "rerun", j["rerun_reads"])'
for lib in "" bwbble_amd/tools_exp/libbwbble_hip_spec.so; do
  echo "== lib ${lib:-product}"
  BWB_LIB=${lib:+$PWD/$lib} timeout 1500 python bench.py --steps 6 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | python3 -c "$show"
done 2>&1 | tee gpurun_out/r2p13/ab.log
BWB_LIB=$PWD/bwbble_amd/tools_exp/libbwbble_hip_spec.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
