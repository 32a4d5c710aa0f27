#!/bin/bash
# The rocprofv3 passes behind profiles/r2_*: run on the GPU box from the repo root (gpurun -- 'bash tools/r2_profile.sh').
#   0. the GPU test suite
#   1. the default bench line (C3: builds genome/index/reads once, cached in /tmp/bwb_bench) and the line with the driver's
#      arguments (--gpus 1 --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1 --no-extras` (same workload, no CPU baseline legs)
#   3. FETCH_SIZE and WRITE_SIZE of `bench.py --steps 2 --warmup 0 --no-extras`, one --pmc pass each (no trace domains)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2_profile; rm -rf $OUT; mkdir -p $OUT
( cd $R && time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/gputests.log 2>&1
tail -3 $OUT/gputests.log
( cd $R && timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
( time timeout 2400 python3 $R/bench.py ) > $OUT/bench_default.log 2>&1
grep '^{"metric"' $OUT/bench_default.log > $OUT/r2_bench_line.json
( time timeout 2400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_driver_args.log 2>&1
grep '^{"metric"' $OUT/bench_driver_args.log > $OUT/r2_bench_line_driver_args.json
tail -4 $OUT/bench_driver_args.log | cut -c1-300
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $OUT/r2_bench_line_under_rocprof.json 2> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1500 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o run -- python3 $R/bench.py --steps 2 --warmup 0 --no-extras > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
done
python3 $R/tools/r2_profile_summary.py $OUT $OUT/profiles
ls -la $OUT/profiles
