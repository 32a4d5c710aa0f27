#!/bin/bash
# Round 5, session 8: parity tests; A/B at C3: s7 (session 7's product) | condpf (HEAD with the heap-top and record prefetches under `if (mask)` as
# before) | product (HEAD: visit counts from the pair's flags, lazy switch of the cached bucket, those two prefetches unconditional)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s8; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s8 "--steps 6 --warmup 2 --no-extras" s7:bwbble_amd/tools_exp/libbwbble_hip_s7.so condpf:bwbble_amd/tools_exp/libbwbble_hip_condpf.so product s7b:bwbble_amd/tools_exp/libbwbble_hip_s7.so condpf2:bwbble_amd/tools_exp/libbwbble_hip_condpf.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
