#!/bin/bash
# Developer tool: HBM traffic of the alignment kernels for bench.py's workload, from rocprofv3 --pmc passes (counters only: no trace
# domains next to --pmc), plus the calibration of how the L2's memory-side read counters tally this kernel's request shapes.
#   tools/pmc_traffic.sh <tag> [bench.py args ...]      e.g.  tools/pmc_traffic.sh r3_c3          (default workload, config C3)
#                                                             tools/pmc_traffic.sh r3_c5 --config C5
# Passes (each its own process, python3 bench.py directly after `--`):
#   0. calibration: tools_exp/gather_bench 7168 7168 meta under TCC_EA0_RDREQ / _32B / TCC_BUBBLE / _DRAM: known numbers of 128-byte
#      bucket requests (cooperative gather) and of 8- and 16-byte per-lane loads (the search kernel's metadata shape)
#   1. bench.py --steps $PMC_STEPS (8) --warmup 0 --no-extras under the same four read counters
#   2. the same under WRITE_SIZE + TCC_EA0_WRREQ / _64B
# tools/pmc_traffic_summary.py turns the CSVs into profiles/<tag>_pmc.json (with the hash of the kernel sources: bench.py quotes the
# file only for the code it was measured on).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
[ -f "$R/bench.py" ] || { echo "pmc_traffic.sh: $R/bench.py not found"; exit 2; }
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum"
make -s -C $R/bwbble_amd gather_bench > /dev/null 2>&1
timeout 600 rocprofv3 --pmc $RD --output-format csv -d $OUT/calib -o run -- $R/bwbble_amd/tools_exp/gather_bench 7168 7168 meta > $OUT/calib.log 2>&1 || echo "calibration pass failed"
timeout 1800 python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-extras > /dev/null 2>&1   # builds genome/index/reads once (cached in /tmp/bwb_bench)
# (round 6) PMC_STEPS steps per pass (default 8: eight slices and the draining launch), and the library's own launch log next to the counters
# (BWB_LAUNCH_LOG: buckets, heap entries and records of every launch), so that the summary prices every dispatch on its own and
# bench.py can put a run's traffic together as (its slices) x (a slice's bytes) + (its draining launches) x (a drain's bytes)
STEPS=${PMC_STEPS:-8}
export BWB_LAUNCH_LOG=$OUT/rd_launches.jsonl
timeout 2400 rocprofv3 --pmc $RD --output-format csv -d $OUT/rd -o run -- python3 $R/bench.py "$@" --steps $STEPS --warmup 0 --no-extras > $OUT/rd.json 2> $OUT/rd.log || echo "read pass failed"
export BWB_LAUNCH_LOG=$OUT/wr_launches.jsonl
timeout 2400 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/wr -o run -- python3 $R/bench.py "$@" --steps $STEPS --warmup 0 --no-extras > $OUT/wr.json 2> $OUT/wr.log || echo "write pass failed"
unset BWB_LAUNCH_LOG
python3 $R/tools/pmc_traffic_summary.py $OUT $R/gpurun_out/${TAG}_pmc.json $STEPS
