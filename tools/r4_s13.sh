#!/bin/bash
# Round-4 session 13 (after the evidence session, same kernel sources): (1) the CPU baseline stated both ways - the reference as it allocates
# its index, and under an interleaving memory policy (oracle/interleave_exec: the boxes have no numactl) - in a default bench.py run;
# (2) config C5 with the DRIVER's arguments (the evidence session used --steps 10 --warmup 2); (3) the event histogram of the shipped
# kl_search (make hist), for the next round's loop work.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 1500 python3 $R/bench.py --steps 3 --warmup 1 ) > $O/bench_default.log 2> $O/bench_default.err
grep '^{"metric"' $O/bench_default.log > $O/r4_bench_line_default_args.json; python3 $R/tools/ab_show.py c3_default < $O/r4_bench_line_default_args.json
python3 - <<PY
import json
d = json.loads(open("$O/r4_bench_line_default_args.json").read())
c = d["cpu_baseline"]
print("cpu_baseline plain:", c["value"], "reads/s at -t", c["threads"], "| interleaved:", c.get("interleaved") if not isinstance(c.get("interleaved"), dict) else (c["interleaved"]["value"], c["interleaved"]["threads"], c["interleaved"]["command"]), "| nodes", c.get("numa_nodes"))
PY
( time timeout 1500 python3 $R/bench.py --config C5 --gpus 1 --steps 20 --warmup 5 ) > $O/r4_bench_line_c5.json 2> $O/c5.err; python3 $R/tools/ab_show.py c5_driver_args < $O/r4_bench_line_c5.json
AB_TIMEOUT=600 bash $R/tools/ab_bench.sh r4s13_hist "--steps 4 --warmup 1 --no-extras" hist:bwbble_amd/tools_exp/libbwbble_hip_hist.so
cp $R/gpurun_out/r4s13_hist/hist.hist $O/ 2>/dev/null; head -c 3000 $O/hist.hist
