#!/bin/bash
# Round 5, session 5: parity tests on the ABI-3 kernels (eight gap runs) with kl_calc_d's buffered list stores; A/B at C3: product | nokq (the
# alignment parameters NOT opaque: the compiler re-reads them from the kernarg segment) | nu16 (16 compacted U rows per gather round) | nt
# (non-temporal bucket loads); basic-block profile at three blocks per CU is not possible (the instrumented kernel needs more registers).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
cd $R
O=$R/gpurun_out/r5s5; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_zz_grch37.py --deselect tests/test_gpu_fullsize.py ) > $O/tests.txt 2>&1
tail -5 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r5s5 "--steps 6 --warmup 2 --no-extras" product nokq:bwbble_amd/tools_exp/libbwbble_hip_nokq.so nu16:bwbble_amd/tools_exp/libbwbble_hip_nu16.so nt:bwbble_amd/tools_exp/libbwbble_hip_nt.so product2 nokq2:bwbble_amd/tools_exp/libbwbble_hip_nokq.so > $O/ab.txt 2>&1
cat $O/ab.txt
python3 $R/bench.py --steps 2 --warmup 0 --ndiff 0 --no-extras 2>/dev/null | python3 $R/tools/ab_show.py n0
