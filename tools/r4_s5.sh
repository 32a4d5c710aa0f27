#!/bin/bash
# Round-4 session 5: kernel arguments of the hot path as opaque scalars (no kernarg reloads), kl_calc_d with the early interval prefetch,
# and the leaner gather as a variant: GPU tests (GRCh37-size file skipped), A/B at C3 against session 3's kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s5; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -40 $O/pytest.log; exit 1; }
BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_g2.so timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_parity.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s5_ab "--steps 6 --warmup 2 --no-extras" s3:_exp/lib_r4s3.so product g2:bwbble_amd/tools_exp/libbwbble_hip_g2.so
