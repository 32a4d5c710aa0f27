#!/bin/bash
# Round-6 evidence session: everything under profiles/r6_* that is quoted for the FINAL kernels comes from this one session on one box.
#   1. the whole GPU suite (with the GRCh37-size parity tests: they leave the C3 index in /tmp/bwb_bench for the steps below) + smoke()
#   2. PMC traffic at C3 with the library's launch log (tools/pmc_traffic.sh r6_c3: 8 steps per pass, every dispatch priced on its own,
#      slices and the draining launch apart) -> profiles/r6_c3_pmc.json, stamped with the hash of the kernel sources
#   3. the bench line with the driver's arguments and all extras; the CLI once more with its .aln checked in two places (tools/cli_check.py)
#   4. the same command under rocprofv3 --kernel-trace --stats (3 steps) -> r6_c3_kernel_stats.csv, r6_c3_kernel_launches.json + the line it printed
#   5. SQ counters of kl_search (tools/pmc_mem.sh, the SQ groups)
#   6. config C5: bench line with the driver's arguments (+ PMC traffic when the clock allows); config C2: bench line
set -u
T0=$(date +%s); LIMIT=${R6_LIMIT_S:-5000}   # the GPU budget left is what bounds this session: optional steps check the clock
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6final; mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -x -q -s ) > $O/pytest.log 2>&1; PRC=$?; echo "pytest exit $PRC"; grep -h "grch37\|passed\|failed\|skipped\|real" $O/pytest.log | tail -12
if [ $PRC -ne 0 ]; then tail -40 $O/pytest.log; echo "GPU tests failed: no evidence is taken on kernels that are not green"; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh r6_c3 > $O/pmc_c3.log 2>&1; tail -4 $O/pmc_c3.log; cp $R/gpurun_out/r6_c3_pmc.json $R/profiles/r6_c3_pmc.json 2>/dev/null   # (so that the bench line below can quote it)
( export BWB_BENCH_BUDGET_S=5000; time timeout 3000 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r6_bench_line_driver_args.json; tail -3 $O/bench_driver_args.err | cut -c1-200
python3 $R/tools/ab_show.py c3_driver_args < $O/r6_bench_line_driver_args.json
( time timeout 900 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/r6_cli_c3.txt 2>&1; tail -7 $O/r6_cli_c3.txt
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/r6_bench_line_under_rocprof.json 2> $O/trace.log
cp $O/trace/run_kernel_stats.csv $O/r6_c3_kernel_stats.csv 2>/dev/null || cp $O/trace/*/run_kernel_stats.csv $O/r6_c3_kernel_stats.csv; head -4 $O/r6_c3_kernel_stats.csv | cut -c1-220
T=$(ls $O/trace/run_kernel_trace.csv $O/trace/*/run_kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/kernel_launches.py $T $O/r6_bench_line_under_rocprof.json $O/r6_c3_kernel_launches.json "every launch of the alignment kernels in rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-extras (tools/r6_final.sh); durations from the kernel trace, in launch order"
rm -rf $O/trace
[ $(left) -gt 2200 ] && { PMC_SETS="1 5 6" bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; sed -n '/^ms /,$p' $O/pmc_mem_c3.log | head -34; }
[ $(left) -gt 1300 ] && { timeout 2400 python3 $R/bench.py --config C5 --steps 20 --warmup 5 --no-extras > $O/r6_bench_line_c5.json 2> $O/c5.err; python3 $R/tools/ab_show.py c5 < $O/r6_bench_line_c5.json; }
[ $(left) -gt 600 ] && { timeout 1200 python3 $R/bench.py --config C2 --steps 20 --warmup 5 --no-extras > $O/r6_bench_line_c2.json 2> $O/c2.err; python3 $R/tools/ab_show.py c2 < $O/r6_bench_line_c2.json; }
[ $(left) -gt 900 ] && { PMC_STEPS=4 bash $R/tools/pmc_traffic.sh r6_c5 --config C5 > $O/pmc_c5.log 2>&1; tail -3 $O/pmc_c5.log; cp $R/gpurun_out/r6_c5_pmc.json $R/profiles/r6_c5_pmc.json 2>/dev/null; }
ls $O
