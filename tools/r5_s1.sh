#!/bin/bash
# Round 5, session 1: (1) the product (per-lane statistics accumulators) on the 6-step C3 metric, (2) the basic-block profile of the
# shipped kernels at C3 (tools/bbprof.py), (3) a PC-sampling trial (beta: last, under its own timeout).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r5s1; mkdir -p $O
bash $R/tools/ab_bench.sh r5s1 "--steps 6 --warmup 2 --no-extras" product > $O/ab.txt 2>&1
# bbprof: sanity first on a small step, then 3 steps of 1 M reads (the counters are totals since reset_stats)
rm -f $O/bb_counts.json
( export BWB_LIB=$R/bwbble_amd/tools_exp/libbwbble_hip_bbprof.so BWB_BBPROF_OUT=$O/bb_counts.json
  timeout 900 python3 $R/bench.py --steps 3 --warmup 0 --reads 1000000 --no-extras > $O/bb_bench.json 2> $O/bb_bench.err )
echo "bbprof rc $?" >> $O/ab.txt
python3 $R/tools/bbprof.py report $O/bb_counts.json > $O/bb_report.txt 2>&1
tail -c 600 $O/bb_bench.json >> $O/ab.txt
# PC sampling (stochastic), the product library, small workload first
for method in stochastic host_trap; do
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $([ $method = stochastic ] && echo cycles || echo time) --pc-sampling-interval $([ $method = stochastic ] && echo 1048576 || echo 1000) \
     --output-format csv -d $O/pcs_$method -o run -- python3 $R/tools/prof_bench.py 3100 10000000 500000 3 2 > $O/pcs_$method.log 2>&1
  echo "pcs $method rc $?" >> $O/ab.txt
  ls -la $O/pcs_$method 2>/dev/null | head >> $O/ab.txt
  find $O/pcs_$method -name "*.csv" -size +60M -exec sh -c 'head -c 50000000 "$1" > "$1.head"; rm "$1"' _ {} \;
done
cat $O/ab.txt
