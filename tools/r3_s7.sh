#!/bin/bash
# round-3 session 7: single-pass rank, blocks-per-CU rule, C5 defaults: tests, C3 and C5 lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s7; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
if grep -q "failed\|error" $O/gputests.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r3s7 "--steps 20 --warmup 5 --no-extras" product20
bash $R/tools/ab_bench.sh r3s7 "--config C5 --steps 8 --warmup 2 --no-extras" c5
