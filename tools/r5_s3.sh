#!/bin/bash
# Round 5, session 3: parity tests on the flattened kernels (+ penalties above 63, fuzz sweep), A/B at C3: r4 | s2 (session 2's product) |
# product (flat dispatch, peeled match loop) | age1 / age2 (wave priority for long-running reads), basic-block profile, 20-step figure.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s3; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s3 "--steps 6 --warmup 2 --no-extras" s2:bwbble_amd/tools_exp/libbwbble_hip_s2.so product age2:bwbble_amd/tools_exp/libbwbble_hip_age2.so age1:bwbble_amd/tools_exp/libbwbble_hip_age1.so r4:bwbble_amd/tools_exp/libbwbble_hip_r4.so s2b:bwbble_amd/tools_exp/libbwbble_hip_s2.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
rm -f $O/bb_counts.json
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err )
echo "bbprof rc $?"
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1
head -8 $O/bb_report.txt
bash $R/tools/ab_bench.sh r5s3_20 "--steps 20 --warmup 2 --no-extras" product age2:bwbble_amd/tools_exp/libbwbble_hip_age2.so > $O/ab20.txt 2>&1
cat $O/ab20.txt
