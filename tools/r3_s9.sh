#!/bin/bash
# round-3 session 9: counters fixed (GPU tests), A/B: side buckets' states fetched only when the expansion pushes to them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s9; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
if grep -q "failed\|error" $O/gputests.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_condbst.so timeout 900 python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_edge_parity.py -m gpu -x -q 2>&1 | tail -2
bash $R/tools/ab_bench.sh r3s9 "--steps 12 --warmup 4 --no-extras" product condbst:bwbble_amd/tools_exp/libbwbble_hip_condbst.so
