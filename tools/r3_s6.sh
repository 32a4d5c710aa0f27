#!/bin/bash
# round-3 session 6: the default bench line with the driver's arguments (all extras), SQ/TLB/L2 counters of kl_search at C3, traffic passes, C5 variants
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 2400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_args.log 2> $O/bench_driver_args.err
grep '^{"metric"' $O/bench_driver_args.log > $O/r3_bench_line_driver_args.json; tail -3 $O/bench_driver_args.err
python3 - <<PY
import json
j=json.loads(open("$O/r3_bench_line_driver_args.json").read())
print("value", j["value"], "frac", j["roofline"]["frac"], "cpu", json.dumps(j.get("cpu_baseline"))[:900])
print("e2e", j.get("end_to_end"), "n0", j.get("also"), "micro", j.get("rank_micro",{}).get("lane_layout"))
PY
bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; tail -60 $O/pmc_mem_c3.log
bash $R/tools/pmc_traffic.sh r3_c3 > $O/pmc.log 2>&1; grep -A12 '"kl_search"' $O/pmc.log | head -30
bash $R/tools/ab_bench.sh r3s6 "--config C5 --steps 12 --warmup 4 --no-extras" "c5_bpc2::BWB_BLOCKS_PER_CU=2"
bash $R/tools/ab_bench.sh r3s6 "--config C5 --reads 2000000 --pool 8000000 --steps 8 --warmup 2 --no-extras" c5_2M "c5_2M_bpc2::BWB_BLOCKS_PER_CU=2"
