#!/bin/bash
# r2 probe 8: s_memtime stamps of kl_search's segments at C3 (diagnostic build `make stamps`), 2 steps.
set -u
mkdir -p gpurun_out/r2p8
BWB_DEBUG=1 BWB_LIB=$PWD/bwbble_amd/tools_exp/libbwbble_hip_stamps.so timeout 1800 python bench.py --steps 3 --warmup 0 --no-extras > gpurun_out/r2p8/stamps.log 2>&1
grep -E "stamps|iterations|kl_search class" gpurun_out/r2p8/stamps.log | cut -c1-700
