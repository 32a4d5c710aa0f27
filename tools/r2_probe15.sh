#!/bin/bash
# r2 probe 15: children staged as 32-bit values relative to a base-table row: GPU tests, then the C3 bench (6 steps) against probe 14's 213.9 k reads/s.
set -u
mkdir -p gpurun_out/r2p15
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2p15/gputests.log 2>&1
tail -12 gpurun_out/r2p15/gputests.log
show='
import sys, json
j = json.loads(sys.stdin.read()); k = j["roofline"]["kernels"]
print("value", j["value"], "ms/step", j["ms_per_step"], "search ms/launch", k["kl_search"]["ms_per_launch"], "launches", k["kl_search"]["launches"], "frac", k["kl_search"]["frac"], "lanes", j["roofline"]["lanes_busy_of_64"], "calc_d ms", k["kl_calc_d"]["ms_per_launch"], "rerun", j["rerun_reads"])'
timeout 1500 python bench.py --steps 6 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | tee gpurun_out/r2p15/c3.json | python3 -c "$show"
