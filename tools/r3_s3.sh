#!/bin/bash
# round-3 session 3: rank from LDS / relative children / three waves per SIMD: GPU tests, then C3 A/B: 3 blocks per CU against 2 (same binary)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3s3
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $R/gpurun_out/r3s3/gputests.log 2>&1; tail -15 $R/gpurun_out/r3s3/gputests.log
if grep -q "failed\|error" $R/gpurun_out/r3s3/gputests.log; then exit 1; fi
bash $R/tools/ab_bench.sh r3s3 "--steps 4 --warmup 1 --no-extras" product "two_blocks::BWB_BLOCKS_PER_CU=2,BWB_CALCD_BLOCKS_PER_CU=2" hist:bwbble_amd/tools_exp/libbwbble_hip_hist.so
bash $R/tools/ab_bench.sh r3s3 "--steps 20 --warmup 5 --no-extras" product20
