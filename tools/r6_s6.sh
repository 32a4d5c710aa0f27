#!/bin/bash
# Round 6, session 6 (product unchanged): can the reads that make the draining launch be named before the search?  tools/predictor_probe2.py at C3 and C5.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r6s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 1200 python3 $R/bench.py --steps 1 --warmup 0 --no-extras > $O/setup.json 2> $O/setup.err   # builds genome / index / reads (cached in /tmp/bwb_bench)
BWB_DEBUG_ITERS=1 timeout 900 python3 $R/tools/predictor_probe2.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 600000 -n 3 > $O/probe_c3.txt 2>&1
cat $O/probe_c3.txt
