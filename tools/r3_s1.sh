#!/bin/bash
# round-3 session 1: GPU tests with the publish fence, list of PMC counters, C3 baseline + event histogram + stamps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3s1
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $R/gpurun_out/r3s1/gputests.log 2>&1; tail -3 $R/gpurun_out/r3s1/gputests.log
rocprofv3 -L > $R/gpurun_out/r3s1/counters.txt 2>&1; wc -l $R/gpurun_out/r3s1/counters.txt
bash $R/tools/ab_bench.sh r3s1 "--steps 4 --warmup 1 --no-extras" product hist:bwbble_amd/tools_exp/libbwbble_hip_hist.so stamps:bwbble_amd/tools_exp/libbwbble_hip_stamps.so
cat $R/gpurun_out/r3s1/hist.hist
