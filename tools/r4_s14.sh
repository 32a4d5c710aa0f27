#!/bin/bash
# Round-4 session 14 (the round's last GPU minutes; the product's kernel sources are NOT changed): A/B of a variant built from a copy of the
# sources - the two per-iteration statistics (heap entries stored / fetched) kept in two registers per lane and summed over the wave once per
# launch, instead of one 64-lane LDS atomic per wave iteration (which is what the new "LDS bank conflicts" of the shipped kernel's SQ
# counters are: 64 lanes on one address).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s14; mkdir -p $O
cd $R
BWB_LIB=$R/_exp/lib_r4s14_accstats.so timeout 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_parity.py -m gpu -x -q > $O/parity_accstats.log 2>&1; echo "parity accstats: exit $? $(tail -1 $O/parity_accstats.log)"
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=400 bash $R/tools/ab_bench.sh r4s14_ab "--steps 6 --warmup 2 --no-extras" product accstats:_exp/lib_r4s14_accstats.so
