#!/bin/bash
# Round 5, session 4: parity tests; A/B at C3: s3 (session 3's product) | product (alignment parameters as opaque scalars: no kernarg reloads in
# the loop; kl_calc_d reads its lists four intervals at a time) | gp1 / gp2 (raised wave priority from section B / A until the gather is issued);
# basic-block profile; the best of them with the driver's 20 steps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s4; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s4 "--steps 6 --warmup 2 --no-extras" s3:bwbble_amd/tools_exp/libbwbble_hip_s3.so product gp1:bwbble_amd/tools_exp/libbwbble_hip_gp1.so gp2:bwbble_amd/tools_exp/libbwbble_hip_gp2.so s3b:bwbble_amd/tools_exp/libbwbble_hip_s3.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
rm -f $O/bb_counts.json
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err )
echo "bbprof rc $?"
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1
head -5 $O/bb_report.txt; grep -A3 "kl_calc_d" $O/bb_report.txt | head -6
bash $R/tools/ab_bench.sh r5s4_20 "--steps 20 --warmup 2 --no-extras" product gp1:bwbble_amd/tools_exp/libbwbble_hip_gp1.so > $O/ab20.txt 2>&1
cat $O/ab20.txt
