#!/bin/bash
# round-3 session 12: the bench line with the driver's arguments once more, with roofline.traffic scaled per step (bench.py fix)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 2400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r3_bench_line_driver_args.json; tail -4 $O/bench_driver_args.err | cut -c1-200
python3 $R/tools/ab_show.py c3_driver_args < $O/r3_bench_line_driver_args.json
