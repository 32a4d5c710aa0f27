#!/bin/bash
# Round-4 session 8: cheaper gather addressing (exec masks kept), heap-entry counters through one LDS atomic:
# GPU tests (GRCh37-size file skipped), then A/B at C3 against session 3 and session 6 kernels.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s8; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -60 $O/pytest.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s8_ab "--steps 6 --warmup 2 --no-extras" s3:_exp/lib_r4s3.so s6:_exp/lib_r4s6.so product
