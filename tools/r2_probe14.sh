#!/bin/bash
# r2 probe 14: (a) TLB / L1 / L2 / SQ counters of kl_search at C3 (tools/pmc_mem.sh); (b) A/B: private chunk runs of a block
# interleaved chunk-major (tools_exp/libbwbble_hip_ilv.so) against the product.
set -u
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r2p14
bash tools/pmc_mem.sh 3100 10000000 2500000 3 > gpurun_out/r2p14/pmc_c3.log 2>&1
cat gpurun_out/r2p14/pmc_c3.log | tail -30
cd $GRAFT_REPO_ROOT
show='
import sys, json
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernels"]
print("value", j["value"], "ms/step", j["ms_per_step"], "search ms/launch", k["kl_search"]["ms_per_launch"], "launches", k["kl_search"]["launches"], "frac", k["kl_search"]["frac"], "lanes", j["roofline"]["lanes_busy_of_64"], "rerun", j["rerun_reads"])'
for lib in bwbble_amd/tools_exp/libbwbble_hip_ilv.so ""; do
  echo "== lib ${lib:-product}"
  BWB_LIB=${lib:+$PWD/$lib} timeout 1500 python bench.py --steps 6 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | python3 -c "$show"
done 2>&1 | tee gpurun_out/r2p14/ab.log
