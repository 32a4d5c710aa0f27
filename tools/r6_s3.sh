#!/bin/bash
# Round 6, session 3: the calculate_d table (kl_calc_d starts from the state after the first 12 steps, looked up by the last 12 bases), three
# blocks per CU as the rule, the CLI's start-up both ways.  Parity tests; A/B at C3 and on the -n 0 path: nodtab (BWB_DTAB=0) | product; the
# table's own build line; the CLI with its start-up timeline, asynchronous and synchronous context creation; C5; the GRCh37-size parity tests
# (the table is active there: chunks of 200 000 reads).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s3; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r6s3 "--steps 6 --warmup 2 --no-extras" nodtab::BWB_DTAB=0 product nodtab2::BWB_DTAB=0 product2 > $O/ab.txt 2>&1
cat $O/ab.txt
BWB_DEBUG=1 timeout 600 python3 $R/bench.py --steps 2 --warmup 0 --no-extras > $O/dbg.json 2> $O/dbg.err; grep "calculate_d table\|chunk pool\|kl_calc_d" $O/dbg.err | head -6 | cut -c1-250
bash $R/tools/ab_bench.sh r6s3n0 "--ndiff 0 --steps 6 --warmup 2 --no-extras" n0_nodtab::BWB_DTAB=0 n0_product > $O/ab_n0.txt 2>&1
cat $O/ab_n0.txt
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
for v in async_dbg sync_dbg async sync async2; do
  ( [ ${v%%_*} = sync ] && export BWB_SYNC_CREATE=1; [ ${v#*_} = dbg ] && export BWB_DEBUG=1
    /usr/bin/time -f "%e s process wall" $R/bwbble_amd/bin/bwbble align -n 3 $FA $FQ /tmp/cli_$v.aln > $O/cli_$v.out 2> $O/cli_$v.err )
  echo "== $v"; grep "^GPUs\|^start-up" $O/cli_$v.out | cut -c1-330; grep "process wall\|bwb host\] worker\|chunk pool\|calculate_d table" $O/cli_$v.err | grep -v "chunk [0-9]* (" | head -8 | cut -c1-250
done
cmp /tmp/cli_async.aln /tmp/cli_sync.aln && echo "async and sync .aln identical"; rm -f /tmp/cli_*.aln
bash $R/tools/ab_bench.sh r6s3c5 "--config C5 --steps 6 --warmup 2 --no-extras" c5_nodtab::BWB_DTAB=0 c5_product > $O/ab_c5.txt 2>&1
cat $O/ab_c5.txt
cd $R
( time timeout 1500 python3 -m pytest tests/test_gpu_zz_grch37.py -m gpu -x -q -s ) > $O/grch37.txt 2>&1; grep -h "grch37\|passed\|failed\|real\|Error\|assert" $O/grch37.txt | tail -12
