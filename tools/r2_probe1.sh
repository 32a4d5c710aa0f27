#!/bin/bash
# r2 probe 1: GPU tests with the slot/slice pipeline, a chr21-scale bench (quick), then the default (C3) bench line.
set -u
mkdir -p gpurun_out/r2p1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2p1/gputests.log 2>&1
tail -15 gpurun_out/r2p1/gputests.log
( time timeout 600 python bench.py --genome-mb 48 --pool 4000000 --reads 1000000 --steps 4 --warmup 1 ) > gpurun_out/r2p1/c2.log 2>&1
tail -5 gpurun_out/r2p1/c2.log | cut -c1-6000
( time BWB_DEBUG=1 timeout 2000 python bench.py --steps 4 --warmup 1 ) > gpurun_out/r2p1/c3.log 2>&1
grep -vE "^\s*$" gpurun_out/r2p1/c3.log | tail -60 | cut -c1-6000
