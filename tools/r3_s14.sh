#!/bin/bash
# round-3 session 14 (the round's last GPU minutes): the GPU tests and smoke() on the final tree, then - if time is left - the chr21-scale line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s14; mkdir -p $O
( cd $R && timeout 400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3; python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/tests_and_smoke.txt
cd /tmp && export TMPDIR=/tmp
timeout 170 python3 $R/bench.py --config C2 --steps 20 --warmup 5 > $O/r3_bench_line_c2.json 2> $O/c2.err; python3 $R/tools/ab_show.py c2 < $O/r3_bench_line_c2.json
