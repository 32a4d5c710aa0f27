#!/bin/bash
# Round 6, session 7: the cooperative exact tail of the draining instantiation (bwb_lane.h COOP: in a launch that drains, the wave's idle
# lanes each take one of the following intervals of ONE lane's exact-tail list through the same gather; the owner appends their children in
# list order).  Parity tests (every one-batch call is a draining launch); A/B at C3: nocoop (BWB_NO_COOP=1) | product, and the launches one by
# one (BWB_DEBUG); the CLI both ways; the GRCh37-size parity tests.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s7; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r6s7 "--steps 6 --warmup 2 --no-extras" nocoop::BWB_NO_COOP=1 product nocoop2::BWB_NO_COOP=1 product2 > $O/ab.txt 2>&1
cat $O/ab.txt
for v in nocoop product; do
  ( [ $v = nocoop ] && export BWB_NO_COOP=1; BWB_DEBUG=1 timeout 600 python3 $R/bench.py --steps 3 --warmup 0 --no-extras > $O/drain_$v.json 2> $O/drain_$v.err )
  echo "== $v"; grep "kl_search class" $O/drain_$v.err | sed -e 's/, pool chunks.*//' | tail -4
done
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
for v in nocoop product; do
  ( [ $v = nocoop ] && export BWB_NO_COOP=1; $R/bwbble_amd/bin/bwbble align -n 3 $FA $FQ /tmp/cli_$v.aln > $O/cli_$v.out 2> $O/cli_$v.err )
  echo "== CLI $v"; grep "^GPUs\|^start-up" $O/cli_$v.out | cut -c1-330
done
cmp /tmp/cli_nocoop.aln /tmp/cli_product.aln && echo "CLI .aln identical with and without the cooperative drain"; rm -f /tmp/cli_*.aln
bash $R/tools/ab_bench.sh r6s7c5 "--config C5 --steps 6 --warmup 2 --no-extras" c5_nocoop::BWB_NO_COOP=1 c5_product > $O/ab_c5.txt 2>&1
cat $O/ab_c5.txt
cd $R
( time timeout 1500 python3 -m pytest tests/test_gpu_zz_grch37.py tests/test_gpu_fullsize.py -m gpu -x -q -s ) > $O/grch37.txt 2>&1; grep -h "grch37\|passed\|failed\|real\|Error\|assert" $O/grch37.txt | tail -12
