#!/bin/bash
# Round-4 session 2: GPU tests on the restructured kl_search (side states in registers, packed two-entry heap-top mirror with in-place
# prefetch ahead of the gather, early list-interval prefetch, late record unpack), then A/B at C3 against the cleaned-up round-3 kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s2; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s2_ab "--steps 6 --warmup 2 --no-extras" base:_exp/lib_r4base.so product base2:_exp/lib_r4base.so
