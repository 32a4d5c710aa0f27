#!/bin/bash
# r2 probe 3: cooperative (LDS-staged) bucket gather in both kernels: GPU tests, chr21-scale and C3 bench lines.
set -u
mkdir -p gpurun_out/r2p3
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2p3/gputests.log 2>&1
tail -15 gpurun_out/r2p3/gputests.log
( time timeout 600 python bench.py --genome-mb 48 --pool 4000000 --reads 1000000 --steps 4 --warmup 1 ) > gpurun_out/r2p3/c2.log 2>&1
tail -3 gpurun_out/r2p3/c2.log | cut -c1-5000
( time timeout 2000 python bench.py --steps 4 --warmup 1 ) > gpurun_out/r2p3/c3.log 2>&1
grep -vE "^\s*$" gpurun_out/r2p3/c3.log | tail -8 | cut -c1-7000
