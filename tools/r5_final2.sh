#!/bin/bash
# Round 5, second evidence session: HOST-ONLY change after tools/r5_final.sh (the OpenMP teams of the host stages capped at 32 threads,
# bwb_host.h; the kernel sources and their hash are those of the first session).  Re-measures what the host code is part of: the bench line
# with the driver's arguments and all extras (cli_end_to_end, host_pipeline, aln2sam, n0, long_stream) and the CLI check.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5final2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( export BWB_BENCH_BUDGET_S=5000; time timeout 3000 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r5_bench_line_driver_args.json; tail -3 $O/bench_driver_args.err | cut -c1-200
python3 $R/tools/ab_show.py c3_driver_args < $O/r5_bench_line_driver_args.json
python3 $R/tools/r5_summary.py $O 2>&1 | grep -A1 "cli_end_to_end\|also\|end_to_end" | cut -c1-1800
( time timeout 900 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/r5_cli_c3.txt 2>&1; tail -6 $O/r5_cli_c3.txt
