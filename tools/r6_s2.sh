#!/bin/bash
# Round 6, session 2: combined gap groups (the insertion rides in the deletion group), kl_calc_d queued ahead on a second stream, the CLI's
# asynchronous start-up.  Parity tests; A/B at C3: nocombine | product | ahead1 / ahead2 / ahead2p (BWB_CALCD_AHEAD) | product2; C5 at two and
# three blocks per CU, private runs of 256 / 128 / 64 chunks; the CLI at C3; the GRCh37-size parity tests (C5 stream + re-run paths).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r6s2; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
NC=bwbble_amd/tools_exp/libbwbble_hip_nocombine.so
bash $R/tools/ab_bench.sh r6s2 "--steps 6 --warmup 2 --no-extras" nocombine:$NC product ahead1::BWB_CALCD_AHEAD=1 ahead2::BWB_CALCD_AHEAD=2 ahead2p::BWB_CALCD_AHEAD=2,BWB_CALCD_PRIO=1 product2 > $O/ab.txt 2>&1
cat $O/ab.txt
( time timeout 900 python3 $R/tools/cli_check.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 5000 -n 3 ) > $O/cli_c3.txt 2>&1; tail -8 $O/cli_c3.txt
bash $R/tools/ab_bench.sh r6s2c5 "--config C5 --steps 4 --warmup 1 --no-extras" c5_2 c5_3::BWB_BLOCKS_PER_CU=3 c5_3_k128::BWB_BLOCKS_PER_CU=3,BWB_KEEP=128 c5_3_k64::BWB_BLOCKS_PER_CU=3,BWB_KEEP=64 c5_nc_3:$NC:BWB_BLOCKS_PER_CU=3 > $O/ab_c5.txt 2>&1
cat $O/ab_c5.txt
BWB_BLOCKS_PER_CU=3 BWB_KEEP=64 BWB_DEBUG=1 timeout 900 python3 $R/bench.py --config C5 --steps 3 --warmup 0 --no-extras > $O/c5_dbg3.json 2> $O/c5_dbg3.err
grep "kl_search" $O/c5_dbg3.err | cut -c1-250 | tail -8
cd $R
( time timeout 1500 python3 -m pytest tests/test_gpu_zz_grch37.py -m gpu -x -q -s ) > $O/grch37.txt 2>&1; grep -h "grch37\|passed\|failed\|real\|Error\|assert" $O/grch37.txt | tail -12
