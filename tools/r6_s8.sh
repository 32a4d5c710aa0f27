#!/bin/bash
# Round 6, session 8 (kernel sources unchanged): `bwbble align` leaves without tearing its context down piece by piece (_exit once the .aln is
# closed) - the CLI's GPU tests, smoke, and the process wall of the 10 M-read run both ways (BWB_FULL_TEARDOWN=1: the orderly way).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s8; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1200 python -m pytest tests/test_host_tools.py tests/test_gpu_parity.py tests/test_dropin_binding.py -m gpu -x -q ) > $O/tests.txt 2>&1; tail -4 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 1200 python3 $R/bench.py --steps 1 --warmup 0 --no-extras > $O/setup.json 2> $O/setup.err
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
sleep 8
for v in full fast full2 fast2; do
  T0=$(date +%s%N)
  ( [ ${v%2} = full ] && export BWB_FULL_TEARDOWN=1; $R/bwbble_amd/bin/bwbble align -n 3 $FA $FQ /tmp/cli_$v.aln > $O/cli_$v.out 2> $O/cli_$v.err )
  T1=$(date +%s%N); echo "== $v: process wall $(( (T1 - T0) / 1000000 )) ms"; grep "^GPUs\|^start-up" $O/cli_$v.out | cut -c1-330
done
cmp /tmp/cli_full.aln /tmp/cli_fast.aln && echo ".aln identical"; rm -f /tmp/cli_*.aln
