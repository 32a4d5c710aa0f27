#!/bin/bash
# round-3 session 8: the numbers that go into profiles/: kernel trace + stats, PMC traffic, bench lines (C3 with the driver's arguments and all
# extras, C2, C5), CLI end to end at C3, randomised parity sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 2400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r3_bench_line_driver_args.json; tail -4 $O/bench_driver_args.err | cut -c1-200
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/r3_bench_line_under_rocprof.json 2> $O/trace.log
ls $O/trace/* | head; cp $O/trace/*/run_kernel_stats.csv $O/r3_c3_kernel_stats.csv 2>/dev/null || cp $O/trace/run_kernel_stats.csv $O/r3_c3_kernel_stats.csv
head -5 $O/r3_c3_kernel_stats.csv | cut -c1-200
bash $R/tools/pmc_traffic.sh r3_c3 > $O/pmc.log 2>&1; tail -5 $O/pmc.log
( time timeout 1500 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/cli_c3.txt 2>&1; cat $O/cli_c3.txt | grep -v "^\[" | tail -6
( cd $R && timeout 900 python3 tools/fuzz_parity.py 24 7 ) > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
( cd $R && BWB_SLICE_ITERS=97 timeout 900 python3 tools/fuzz_parity.py 16 8 ) > $O/fuzz_sliced.txt 2>&1; tail -2 $O/fuzz_sliced.txt
timeout 1200 python3 $R/bench.py --config C2 --steps 20 --warmup 5 > $O/r3_bench_line_c2.json 2> $O/c2.err
timeout 2400 python3 $R/bench.py --config C5 --steps 10 --warmup 2 > $O/r3_bench_line_c5.json 2> $O/c5.err
for f in c2 c5; do python3 $R/tools/ab_show.py $f < $O/r3_bench_line_$f.json; done
