#!/bin/bash
# Round-4 session 7: perturbation analysis at C3 - what does the loop pay for 128 more vector / scalar instructions per wave iteration?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s7_ab "--steps 6 --warmup 2 --no-extras" product pvalu:bwbble_amd/tools_exp/libbwbble_hip_pvalu.so psalu:bwbble_amd/tools_exp/libbwbble_hip_psalu.so
