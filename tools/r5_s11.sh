#!/bin/bash
# Round 5, session 11 (product unchanged): what an instruction is worth in the final loop - 128 extra v_nop (pv) / 128 extra s_nop 0 (ps)
# per wave iteration of kl_search against the product, for the next round's choice of where to cut.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
O=$R/gpurun_out/r5s11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s11 "--steps 6 --warmup 2 --no-extras" product pv:bwbble_amd/tools_exp/libbwbble_hip_pv.so ps:bwbble_amd/tools_exp/libbwbble_hip_ps.so product2 > $O/ab.txt 2>&1
cat $O/ab.txt
for b in 2 1; do BWB_BLOCKS_PER_CU=$b timeout 600 python3 $R/bench.py --steps 6 --warmup 2 --no-extras > $O/blocks$b.json 2> $O/blocks$b.err; python3 $R/tools/ab_show.py blocks_per_cu_$b < $O/blocks$b.json; done 2>&1 | tee $O/blocks.txt
