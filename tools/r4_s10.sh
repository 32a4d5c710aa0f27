#!/bin/bash
# Round-4 session 10: session 9 repaired (lane number a plain constant again, two-value slice form, non-hoistable lane id in alloc): GPU tests, A/B at C3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s10; mkdir -p $O
cd $R
( time BWB_SKIP_GRCH37=1 timeout 1200 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "passed\|failed\|skipped\|real" $O/pytest.log | tail -4
grep -q " failed\|error" $O/pytest.log && { tail -60 $O/pytest.log; exit 1; }
BWB_DEBUG=1 timeout 120 python3 -c "
import bwbble_amd as bw, os
ctx = bw.Context('tests/golden/toy.fa.bwt')
s, l = bw.encode_reads(bw.read_fastq('tests/golden/toy.fq', max_reads=50))
ctx.align(bw.params(['-n','3']), s, l)
" 2>&1 | grep "fit a CU"
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s10_ab "--steps 6 --warmup 2 --no-extras" s6:_exp/lib_r4s6.so s8:_exp/lib_r4s8.so s9:_exp/lib_r4s9.so product
