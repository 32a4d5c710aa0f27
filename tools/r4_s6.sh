#!/bin/bash
# Round-4 session 6: the per-position data as 16-byte records of four positions kept in registers (kl_calc_d writes them, kl_search loads
# one when the popped entry leaves the record's positions): GPU tests (GRCh37-size file skipped), then A/B at C3 against session 3's kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s6; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -60 $O/pytest.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s6_ab "--steps 6 --warmup 2 --no-extras" s3:_exp/lib_r4s3.so product
