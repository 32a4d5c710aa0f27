#!/bin/bash
# Round-4 session 1 (prepared at the end of round 3, not yet run): what DESIGN.md section 8 item 0 asks for, in one GPU session at
# GRCh37 scale (config C3).  Run tools/r4_prepare_exp.sh on the CPU first.
#   1. the SQ / LDS / TLB counters of the FINAL kl_search (64-character buckets): tools/pmc_mem.sh
#   2. A/B of the exp-r4 builds against the product on the same inputs: tools/ab_bench.sh (genome, index and reads are cached by step 1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; sed -n '/^ms /,$p' $O/pmc_mem_c3.log | head -60
V=""; for n in laterec lateside g2 combo karg kg kgall; do [ -f $R/bwbble_amd/tools_exp/libbwbble_hip_$n.so ] && V="$V $n:bwbble_amd/tools_exp/libbwbble_hip_$n.so"; done
# (the karg* builds have never run on a GPU: the A/B lines are their first test; run the GPU test suite with BWB_LIB=<lib> before adopting one)
AB_TIMEOUT=600 bash $R/tools/ab_bench.sh r4s1_ab "--steps 6 --warmup 2 --no-extras" product $V product2:
