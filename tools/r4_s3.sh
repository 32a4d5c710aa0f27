#!/bin/bash
# Round-4 session 3: the whole GPU suite (now with the GRCh37-size parity test and the pipelined CLI), then at C3: product against the
# private-run-first allocator, the s_memtime stamps of the restructured loop, and the SQ counters of the new kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4s3; mkdir -p $O
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -x -q -s ) > $O/pytest.log 2>&1; echo "pytest exit $?"; grep -h "grch37\|passed\|failed\|skipped\|real" $O/pytest.log | tail -8
cd /tmp && export TMPDIR=/tmp
AB_TIMEOUT=700 bash $R/tools/ab_bench.sh r4s3_ab "--steps 6 --warmup 2 --no-extras" product privfirst:bwbble_amd/tools_exp/libbwbble_hip_privfirst.so stamps:bwbble_amd/tools_exp/libbwbble_hip_stamps.so
PMC_SETS="5 6 7" bash $R/tools/pmc_mem.sh 3100 10000000 2500000 3 > $O/pmc_mem_c3.log 2>&1; sed -n '/^ms /,$p' $O/pmc_mem_c3.log | head -40
