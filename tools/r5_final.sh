#!/bin/bash
# Round-5 evidence session: everything under profiles/r5_* that is quoted for the FINAL kernels comes from this one session on one box.
#   1. the whole GPU suite (with the GRCh37-size parity tests: they leave the C3 index in /tmp/bwb_bench for the steps below) + smoke()
#   2. PMC traffic at C3 (tools/pmc_traffic.sh r5_c3) -> profiles/r5_c3_pmc.json, stamped with the hash of the kernel sources
#   3. the bench line with the driver's arguments and all extras (cpu_baseline, end_to_end, rank_micro, n0, cli_end_to_end incl. long_stream,
#      n0, host_pipeline, aln2sam); the CLI once more with its .aln checked in two places (tools/cli_check.py)
#   4. the same command under rocprofv3 --kernel-trace --stats (3 steps) -> r5_c3_kernel_stats.csv, r5_c3_kernel_launches.json + the line it printed
#   5. SQ counters of kl_search (tools/pmc_mem.sh, the SQ groups) and the TLB group
#   6. (the basic-block profile of these kernels is session 10's: profiles/r5_bbprof_s10.txt - same sources)
#   7. config C5: PMC traffic + bench line (and once with three blocks per CU forced: what VERDICT r4's item 2 is about); config C2: bench line
set -u
T0=$(date +%s); LIMIT=${R5_LIMIT_S:-5000}   # the GPU budget left is what bounds this session: optional steps check the clock
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5final; mkdir -p $O
cd $R
( time timeout 1800 python3 -m pytest tests -m gpu -x -q -s ) > $O/pytest.log 2>&1; PRC=$?; echo "pytest exit $PRC"; grep -h "grch37\|passed\|failed\|skipped\|real" $O/pytest.log | tail -8
if [ $PRC -ne 0 ]; then tail -40 $O/pytest.log; echo "GPU tests failed: no evidence is taken on kernels that are not green"; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh r5_c3 > $O/pmc_c3.log 2>&1; tail -3 $O/pmc_c3.log; cp $R/gpurun_out/r5_c3_pmc.json $R/profiles/r5_c3_pmc.json 2>/dev/null   # (so that the bench line below can quote it)
( export BWB_BENCH_BUDGET_S=5000; time timeout 3000 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r5_bench_line_driver_args.json; tail -3 $O/bench_driver_args.err | cut -c1-200
python3 $R/tools/ab_show.py c3_driver_args < $O/r5_bench_line_driver_args.json
( time timeout 900 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/r5_cli_c3.txt 2>&1; tail -6 $O/r5_cli_c3.txt
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/r5_bench_line_under_rocprof.json 2> $O/trace.log
cp $O/trace/run_kernel_stats.csv $O/r5_c3_kernel_stats.csv 2>/dev/null || cp $O/trace/*/run_kernel_stats.csv $O/r5_c3_kernel_stats.csv; head -4 $O/r5_c3_kernel_stats.csv | cut -c1-220
T=$(ls $O/trace/run_kernel_trace.csv $O/trace/*/run_kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/kernel_launches.py $T $O/r5_bench_line_under_rocprof.json $O/r5_c3_kernel_launches.json "every launch of the alignment kernels in rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-extras (tools/r5_final.sh); durations from the kernel trace, in launch order"
rm -rf $O/trace
PMC_SETS="1 5 6 7" bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; sed -n '/^ms /,$p' $O/pmc_mem_c3.log | head -34
[ $(left) -gt 1500 ] && bash $R/tools/pmc_traffic.sh r5_c5 --config C5 > $O/pmc_c5.log 2>&1; tail -3 $O/pmc_c5.log; cp $R/gpurun_out/r5_c5_pmc.json $R/profiles/r5_c5_pmc.json 2>/dev/null
[ $(left) -gt 700 ] && timeout 2400 python3 $R/bench.py --config C5 --steps 20 --warmup 5 --no-extras > $O/r5_bench_line_c5.json 2> $O/c5.err; python3 $R/tools/ab_show.py c5 < $O/r5_bench_line_c5.json
[ $(left) -gt 400 ] && BWB_BLOCKS_PER_CU=3 timeout 1500 python3 $R/bench.py --config C5 --steps 10 --warmup 2 --no-extras > $O/r5_bench_line_c5_three_blocks.json 2> $O/c5b3.err; python3 $R/tools/ab_show.py c5_three_blocks < $O/r5_bench_line_c5_three_blocks.json
[ $(left) -gt 300 ] && timeout 1200 python3 $R/bench.py --config C2 --steps 20 --warmup 5 --no-extras > $O/r5_bench_line_c2.json 2> $O/c2.err; python3 $R/tools/ab_show.py c2 < $O/r5_bench_line_c2.json
ls $O
