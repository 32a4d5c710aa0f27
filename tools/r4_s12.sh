#!/bin/bash
# Round-4 session 12: product = session 11's + 24 compacted U rows per round (LDS per block 49.6 KB) + blocks per CU bounded by LDS granules.
# GPU tests; parity subset on the combined experiment build; A/B at C3: s11 | product | b2full (the full second base table, which now
# fits) | gbuf (the gather as structured-buffer loads: 3 instead of 9 instructions a load) | gbufb2 (both); the CLI end to end at C3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s12; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -60 $O/pytest.log; exit 1; }
for v in gbufb2 gbuf b2full; do
  BWB_LIB=$R/_exp/lib_r4s12_$v.so timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_parity.py -m gpu -x -q > $O/parity_$v.log 2>&1; echo "parity $v: exit $? $(tail -1 $O/parity_$v.log)"
  [ $v = gbufb2 ] && ! grep -q " failed\|error" $O/parity_$v.log && break   # (the combination passed: its parts need no run of their own)
done
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s12_ab "--steps 6 --warmup 2 --no-extras" s11:_exp/lib_r4s11.so product b2full:_exp/lib_r4s12_b2full.so gbuf:_exp/lib_r4s12_gbuf.so gbufb2:_exp/lib_r4s12_gbufb2.so
grep -o '"reads_parked_per_step": [0-9]*' $R/gpurun_out/r4s12_ab/*.json
FA=/tmp/bwb_bench/genome_3100000000.fa; FQ=/tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq
ls -la /tmp/bwb_bench | head -20
( time timeout 900 python3 $R/tools/cli_check.py $FA $FQ 3000 -n 3 ) > $O/cli_c3.txt 2>&1; tail -8 $O/cli_c3.txt
