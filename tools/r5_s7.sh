#!/bin/bash
# Round 5, session 7: parity tests; A/B at C3: s5 (session 5's product) | product (record bytes by one v_perm_b32 each, 32-bit gap-run arithmetic in
# the 16-byte-entry kernels, the non-empty mask by compare + add-with-carry chains); product with the driver's 20 steps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s7; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s7 "--steps 6 --warmup 2 --no-extras" s5:bwbble_amd/tools_exp/libbwbble_hip_s5.so product s5b:bwbble_amd/tools_exp/libbwbble_hip_s5.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
bash $R/tools/ab_bench.sh r5s7_20 "--steps 20 --warmup 2 --no-extras" product > $O/ab20.txt 2>&1
cat $O/ab20.txt
