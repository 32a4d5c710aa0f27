#!/bin/bash
# Round 6, session 1: count-only pushes (bwb_lane.h, expansion).  Parity tests; A/B at C3: nophantom (every push stored = round 5) | product;
# event histogram at C3 and C5 (how many pushes are count-only); C5 at two and at three blocks per CU with the pool's high-water mark.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s1; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
NP=bwbble_amd/tools_exp/libbwbble_hip_nophantom.so
HI=bwbble_amd/tools_exp/libbwbble_hip_hist.so
bash $R/tools/ab_bench.sh r6s1 "--steps 6 --warmup 2 --no-extras" nophantom:$NP product nophantom2:$NP product2 > $O/ab.txt 2>&1
cat $O/ab.txt
bash $R/tools/ab_bench.sh r6s1 "--steps 3 --warmup 1 --no-extras" hist_c3:$HI > $O/ab_hist.txt 2>&1
cat $O/r6s1/hist_c3.hist 2>/dev/null; cat $R/gpurun_out/r6s1/hist_c3.hist
# the launches one by one (BWB_DEBUG synchronises after every launch and prints the pool's fill): the drain, the pool
BWB_DEBUG=1 timeout 600 python3 $R/bench.py --steps 3 --warmup 0 --no-extras > $O/drain_product.json 2> $O/drain_product.err
grep "kl_search" $O/drain_product.err | tail -5
# C5
bash $R/tools/ab_bench.sh r6s1c5 "--config C5 --steps 6 --warmup 2 --no-extras" c5_np_2:$NP c5_prod_2 c5_prod_3::BWB_BLOCKS_PER_CU=3 c5_np_3:$NP:BWB_BLOCKS_PER_CU=3 > $O/ab_c5.txt 2>&1
cat $O/ab_c5.txt
BWB_BLOCKS_PER_CU=3 BWB_DEBUG=1 timeout 900 python3 $R/bench.py --config C5 --steps 4 --warmup 0 --no-extras > $O/c5_dbg3.json 2> $O/c5_dbg3.err
grep "kl_search" $O/c5_dbg3.err | tail -6
bash $R/tools/ab_bench.sh r6s1c5 "--config C5 --steps 3 --warmup 1 --no-extras" hist_c5:$HI > $O/ab_hist_c5.txt 2>&1
cat $R/gpurun_out/r6s1c5/hist_c5.hist
