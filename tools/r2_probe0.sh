#!/bin/bash
# r2 baseline probe: GPU tests at HEAD, then C3-scale stage timings with the round-1 bench (2.5 M-read batch).
set -u
mkdir -p gpurun_out/r2p0
free -g | head -2; nproc; df -h /tmp | tail -1; rocm-smi --showmeminfo vram 2>/dev/null | head -5
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2p0/gputests.log 2>&1
tail -3 gpurun_out/r2p0/gputests.log
( time BWB_DEBUG=1 timeout 1500 python bench.py --genome-mb 3100 --reads 2500000 --ndiff 3 --steps 2 --warmup 0 --cpu-sample 500 ) > gpurun_out/r2p0/c3.log 2>&1
grep -vE "^\s*$" gpurun_out/r2p0/c3.log | tail -40 | cut -c1-3000
ls -la /tmp/bwb_bench | head
