#!/bin/bash
# Round 6, session 4: the calculate_d table looked up after every restart too; the chunk pool kept across uploads (session 3 found it given
# back and taken again - five seconds - at every upload).  The table's parity tests; A/B at C3 and on the -n 0 path: nodtab | product; the CLI
# with its start-up timeline, asynchronous and synchronous context creation; C5.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s4; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_edge_parity.py tests/test_gpu_c3_paths.py tests/test_gpu_parity.py -m gpu -x -q -k "table or superblock or calc_d or golden or toy" ) > $O/tests.txt 2>&1
tail -4 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r6s4 "--steps 6 --warmup 2 --no-extras" nodtab::BWB_DTAB=0 product product2 > $O/ab.txt 2>&1
cat $O/ab.txt
bash $R/tools/ab_bench.sh r6s4n0 "--ndiff 0 --steps 6 --warmup 2 --no-extras" n0_nodtab::BWB_DTAB=0 n0_product > $O/ab_n0.txt 2>&1
cat $O/ab_n0.txt
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
for v in async_dbg sync_dbg async sync async2; do
  ( [ ${v%%_*} = sync ] && export BWB_SYNC_CREATE=1; [ ${v#*_} = dbg ] && export BWB_DEBUG=1
    T0=$(date +%s.%N); $R/bwbble_amd/bin/bwbble align -n 3 $FA $FQ /tmp/cli_$v.aln > $O/cli_$v.out 2> $O/cli_$v.err; echo "process wall $(echo "$(date +%s.%N) - $T0" | bc) s" >> $O/cli_$v.err )
  echo "== $v"; grep "^GPUs\|^start-up" $O/cli_$v.out | cut -c1-330; grep "process wall\|bwb host\] worker\|chunk pool\|calculate_d table" $O/cli_$v.err | grep -v "chunk [0-9]* (" | head -8 | cut -c1-250
done
cmp /tmp/cli_async.aln /tmp/cli_sync.aln && echo "async and sync .aln identical"; rm -f /tmp/cli_*.aln
bash $R/tools/ab_bench.sh r6s4c5 "--config C5 --steps 6 --warmup 2 --no-extras" c5_product > $O/ab_c5.txt 2>&1
cat $O/ab_c5.txt
