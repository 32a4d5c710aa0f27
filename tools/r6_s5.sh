#!/bin/bash
# Round 6, session 5 (product unchanged: the kernel sources of the evidence session): what the loop looks like now - the event histogram
# (make hist) and the basic-block profile (tools/bbprof.py) of the shipped kernels at C3.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r6s5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r6s5 "--steps 3 --warmup 1 --no-extras" hist_c3:bwbble_amd/tools_exp/libbwbble_hip_hist.so > $O/ab_hist.txt 2>&1
cat $R/gpurun_out/r6s5/hist_c3.hist
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err ); echo "bbprof rc $?"
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1; head -14 $O/bb_report.txt
