#!/bin/bash
# Round 5, session 2: parity tests (all but the GRCh37-size ones) on the new kernels (side_finish v_bcnt/SDWA, branchless list_add, XOR row
# rotation, kl_calc_d with its own LDS map), A/B at C3 against the round-4 kernel, kl_calc_d at 4 waves / 8 U rows and 3 waves / 16 U rows,
# basic-block profile of the new kernels.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s2; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s2 "--steps 6 --warmup 2 --no-extras" r4:bwbble_amd/tools_exp/libbwbble_hip_r4.so product c8w4:bwbble_amd/tools_exp/libbwbble_hip_c8w4.so c16w3:bwbble_amd/tools_exp/libbwbble_hip_c16w3.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
rm -f $O/bb_counts.json
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err )
echo "bbprof rc $?"
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1
head -12 $O/bb_report.txt
