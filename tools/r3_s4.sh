#!/bin/bash
# round-3 session 4: occupancy clamp + launch log at C3, traffic passes, C5 and C2 lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s4; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -6 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
BWB_DEBUG=1 timeout 1500 python3 $R/bench.py --steps 6 --warmup 0 --no-extras > $O/debug6.json 2> $O/debug6.err; grep "^\[bwb\]" $O/debug6.err | cut -c1-330 | head -40
bash $R/tools/ab_bench.sh r3s4 "--steps 4 --warmup 1 --no-extras" product
bash $R/tools/ab_bench.sh r3s4 "--steps 20 --warmup 5 --no-extras" product20
bash $R/tools/pmc_traffic.sh r3_c3 > $O/pmc.log 2>&1; tail -60 $O/pmc.log | cut -c1-200
bash $R/tools/ab_bench.sh r3s4 "--config C2 --steps 20 --warmup 5 --no-extras" c2_20
bash $R/tools/ab_bench.sh r3s4 "--config C5 --steps 4 --warmup 1 --no-extras" c5
