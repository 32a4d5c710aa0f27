#!/bin/bash
# Round-4 session 4: free-list head refilled in place, first list interval in registers, batched hit-list loads, record prefetched in
# place: GPU tests (the GRCh37-size file is skipped this time: it ran in session 3), then A/B at C3 against session 3's kernel + stamps.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s4; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -40 $O/pytest.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s4_ab "--steps 6 --warmup 2 --no-extras" s3:_exp/lib_r4s3.so product stamps:bwbble_amd/tools_exp/libbwbble_hip_stamps.so
