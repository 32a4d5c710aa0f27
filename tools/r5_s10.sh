#!/bin/bash
# Round 5, session 10: parity tests; A/B at C3: s9 (session 9's product) | product (gather loads without exec masks, prune/allow logic as expressions,
# unconditional tail store, no load pending on the back edge); basic-block profile; product with the driver's 20 steps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s10; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s10 "--steps 6 --warmup 2 --no-extras" s9:bwbble_amd/tools_exp/libbwbble_hip_s9.so product s9b:bwbble_amd/tools_exp/libbwbble_hip_s9.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err ); echo "bbprof rc $?"
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1; head -12 $O/bb_report.txt
bash $R/tools/ab_bench.sh r5s10_20 "--steps 20 --warmup 2 --no-extras" product > $O/ab20.txt 2>&1
cat $O/ab20.txt
# the draining launch on its own (BWB_DEBUG times every launch): s9 against the product
for v in s9 product; do
  L=$R/bwbble_amd/libbwbble_hip.so; [ $v = s9 ] && L=$R/bwbble_amd/tools_exp/libbwbble_hip_s9.so
  BWB_LIB=$L BWB_DEBUG=1 timeout 600 python3 $R/bench.py --steps 3 --warmup 0 --no-extras > $O/drain_$v.json 2> $O/drain_$v.err
  echo "== $v"; grep "kl_search" $O/drain_$v.err | sed -e 's/, pool chunks.*//' | tail -5
done
