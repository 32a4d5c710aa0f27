#!/bin/bash
# round-3 session 13: the lane kernels on 64-character buckets (GPU tests of this code: tools/r3_s13.sh's predecessor on the
# experiment branch, 89 passed): PMC traffic re-stamped for the new kernel sources, then the bench line with the driver's arguments
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh r3_c3 > $O/pmc.log 2>&1; tail -3 $O/pmc.log; cp $R/gpurun_out/r3_c3_pmc.json $R/profiles/r3_c3_pmc.json; cp $R/gpurun_out/r3_c3_pmc.json $O/
( time timeout 900 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r3_bench_line_driver_args.json; tail -4 $O/bench_driver_args.err | cut -c1-200
python3 $R/tools/ab_show.py c3_driver_args < $O/r3_bench_line_driver_args.json
