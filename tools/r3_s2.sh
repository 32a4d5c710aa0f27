#!/bin/bash
# round-3 session 2: deletion groups + unstored last match + running num_best: GPU tests, then C3 A/B (product, hist)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3s2
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $R/gpurun_out/r3s2/gputests.log 2>&1; tail -15 $R/gpurun_out/r3s2/gputests.log
if grep -q "failed" $R/gpurun_out/r3s2/gputests.log; then exit 1; fi
bash $R/tools/ab_bench.sh r3s2 "--steps 4 --warmup 1 --no-extras" product hist:bwbble_amd/tools_exp/libbwbble_hip_hist.so
cat $R/gpurun_out/r3s2/hist.hist
