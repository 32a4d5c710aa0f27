#!/bin/bash
# round-3 session 11: the numbers that go into profiles/ for the final kernels: bench line with the driver's arguments and all extras,
# kernel trace + stats, PMC traffic (hash-stamped), SQ/TLB counters, CLI end to end, C2 and C5 lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s11; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
if grep -q "failed\|error" $O/gputests.log; then exit 1; fi
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh r3_c3 > $O/pmc.log 2>&1; tail -3 $O/pmc.log; cp $R/gpurun_out/r3_c3_pmc.json $R/profiles/r3_c3_pmc.json   # (so that the bench line below can quote it)
( time timeout 2400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r3_bench_line_driver_args.json; tail -4 $O/bench_driver_args.err | cut -c1-200
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/r3_bench_line_under_rocprof.json 2> $O/trace.log
cp $O/trace/run_kernel_stats.csv $O/r3_c3_kernel_stats.csv; head -4 $O/r3_c3_kernel_stats.csv | cut -c1-220
bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; sed -n '/^ms /,$p' $O/pmc_mem_c3.log | head -45
( time timeout 1500 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/cli_c3.txt 2>&1; grep -v "^\[" $O/cli_c3.txt | tail -6
timeout 1200 python3 $R/bench.py --config C2 --steps 20 --warmup 5 > $O/r3_bench_line_c2.json 2> $O/c2.err
timeout 2400 python3 $R/bench.py --config C5 --steps 10 --warmup 2 > $O/r3_bench_line_c5.json 2> $O/c5.err
for f in c2 c5; do python3 $R/tools/ab_show.py $f < $O/r3_bench_line_$f.json; done
python3 $R/tools/ab_show.py c3_driver_args < $O/r3_bench_line_driver_args.json
