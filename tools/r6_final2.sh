#!/bin/bash
# Round 6, second evidence session: HOST-SIDE change of bwb_hip.hip after tools/r6_final.sh (the chunk pool of a `-n 0` run is a few chunks per
# lane; the calculate_d table builds in a temporary allocation when the pool is that small) - the device code is the first session's, the hash
# of the kernel SOURCES is not.  Re-takes what carries the hash or runs the changed path: the affected GPU tests, PMC traffic at C3
# (profiles/r6_c3_pmc.json), the bench line with the driver's arguments and all extras, the kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6final2; mkdir -p $O
cd $R
( time timeout 1500 python3 -m pytest tests/test_gpu_edge_parity.py tests/test_host_tools.py tests/test_gpu_parity.py tests/test_gpu_c3_paths.py -m gpu -x -q ) > $O/pytest.log 2>&1; PRC=$?; echo "pytest exit $PRC"; tail -4 $O/pytest.log
if [ $PRC -ne 0 ]; then tail -40 $O/pytest.log; exit 1; fi
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_traffic.sh r6_c3 > $O/pmc_c3.log 2>&1; tail -3 $O/pmc_c3.log; cp $R/gpurun_out/r6_c3_pmc.json $R/profiles/r6_c3_pmc.json 2>/dev/null
( export BWB_BENCH_BUDGET_S=5000; time timeout 3000 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r6_bench_line_driver_args.json; python3 $R/tools/ab_show.py c3_driver_args < $O/r6_bench_line_driver_args.json
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > $O/r6_bench_line_under_rocprof.json 2> $O/trace.log
cp $O/trace/run_kernel_stats.csv $O/r6_c3_kernel_stats.csv 2>/dev/null || cp $O/trace/*/run_kernel_stats.csv $O/r6_c3_kernel_stats.csv; head -3 $O/r6_c3_kernel_stats.csv | cut -c1-200
T=$(ls $O/trace/run_kernel_trace.csv $O/trace/*/run_kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/kernel_launches.py $T $O/r6_bench_line_under_rocprof.json $O/r6_c3_kernel_launches.json "every launch of the alignment kernels in rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-extras (tools/r6_final2.sh); durations from the kernel trace, in launch order"
rm -rf $O/trace
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
sleep 8; BWB_DEBUG=1 $R/bwbble_amd/bin/bwbble align -n 0 $FA $FQ /tmp/n0.aln > $O/cli_n0.out 2> $O/cli_n0.err; grep "^GPUs\|^start-up" $O/cli_n0.out | cut -c1-300; grep "chunk pool\|calculate_d table" $O/cli_n0.err | head -3
ls $O
