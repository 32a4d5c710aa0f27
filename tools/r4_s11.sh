#!/bin/bash
# Round-4 session 11: what sessions 9-10 really measured (a second base table = 1 280 bytes more LDS = two blocks per CU although the occupancy
# query said three; an LDS atomic with a provably uniform address = a 64-trip scalar loop in every iteration).  Now: one extra base row instead
# of a table, the atomic's index opaque.  LDS probe (how many bytes per block still give three resident blocks), GPU tests, A/B at C3 against
# session 8's build, the full second table (8 rows) and 24 instead of 32 compacted U rows per round.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s11; mkdir -p $O
cd $R
( cd /tmp && hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/lds_probe $R/bwbble_amd/tools_exp/lds_probe.hip && timeout 120 /tmp/lds_probe ) > $O/lds_probe.txt 2>&1; cat $O/lds_probe.txt | grep -v "^5[0-9]* 3 768$"
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -60 $O/pytest.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s11_ab "--steps 6 --warmup 2 --no-extras" s8:_exp/lib_r4s8.so product b2full:_exp/lib_r4s11_b2full.so nu24:_exp/lib_r4s11_nu24.so
grep -o '"reads_parked_per_step": [0-9]*' $R/gpurun_out/r4s11_ab/*.json
