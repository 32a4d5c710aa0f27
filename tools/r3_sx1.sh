#!/bin/bash
# round-3 experiment session (run from the worktree of branch exp-bkt64 at <repo>/_exp, before the merge): GPU tests on the 64-character-bucket
# build, then A/B against the product library of that moment (128-character buckets) at C2 and C3 -> profiles/r3_bkt64_ab.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; X=$R/_exp; O=$R/gpurun_out/x1; mkdir -p $O
cd $X
timeout 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/tests.txt
grep -q " passed" $O/tests.txt && ! grep -q "failed\|error" $O/tests.txt || { echo "tests not green: no bench"; exit 1; }
export GRAFT_REPO_ROOT=$X
AB_TIMEOUT=400 bash $X/tools/ab_bench.sh x1c2 "--config C2 --steps 8 --warmup 2 --no-extras" bkt64 base:../bwbble_amd/libbwbble_hip.so
AB_TIMEOUT=900 bash $X/tools/ab_bench.sh x1c3 "--steps 4 --warmup 1 --no-extras" bkt64 base:../bwbble_amd/libbwbble_hip.so
cp -r $X/gpurun_out/x1c2 $X/gpurun_out/x1c3 $O/ 2>/dev/null
