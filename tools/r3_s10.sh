#!/bin/bash
# round-3 session 10: conditional bucket-state fetch adopted; A/B: per-position records in the lane's scratch against the slot's buffer
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s10; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
if grep -q "failed\|error" $O/gputests.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r3s10 "--steps 12 --warmup 4 --no-extras" product norecs:bwbble_amd/tools_exp/libbwbble_hip_norecs.so
