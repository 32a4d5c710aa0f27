#!/bin/bash
# Developer tool: the rocprofv3 passes behind profiles/<round>_bench_*: kernel trace + stats of the default bench.py
# command, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.  Run on the GPU box from the repo root:
#   bash bwbble_amd/tools/make_profile.sh r1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/profile_$TAG; rm -rf $OUT; mkdir -p $OUT
python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 500 > /dev/null 2>&1   # builds the workload files once
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $R/bench.py --steps 2 --warmup 1 > $OUT/bench_line.json 2> $OUT/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o run -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 500 > /dev/null 2> $OUT/pmc_$c.log
done
mkdir -p $OUT/profiles
python3 $R/bwbble_amd/tools/make_profile.py $OUT $OUT/profiles $TAG
# gpurun merges gpurun_out/ back; copy gpurun_out/profile_$TAG/profiles/* into profiles/ and commit
