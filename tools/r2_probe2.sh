#!/bin/bash
# r2 probe 2: gather shape experiment at 1 GiB / 7 GiB, GPU tests, C3 bench (4 steps, async).
set -u
mkdir -p gpurun_out/r2p2
( for sz in "1024" "7168" "7168 1024"; do timeout 300 bwbble_amd/tools_exp/gather_bench $sz; done ) > gpurun_out/r2p2/gather.log 2>&1
cat gpurun_out/r2p2/gather.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2p2/gputests.log 2>&1
tail -15 gpurun_out/r2p2/gputests.log
( time timeout 2000 python bench.py --steps 4 --warmup 1 ) > gpurun_out/r2p2/c3.log 2>&1
grep -vE "^\s*$" gpurun_out/r2p2/c3.log | tail -12 | cut -c1-7000
