#!/bin/bash
# Round 6, third evidence session (final sources, hash as tools/r6_final2.sh): the whole GPU suite + smoke once more on the tree that ships,
# then config C5: PMC traffic (re-stamped) and the bench line with the driver's arguments; config C2: bench line.
set -u
T0=$(date +%s); LIMIT=${R6_LIMIT_S:-2400}
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6final3; mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -x -q -s ) > $O/pytest.log 2>&1; PRC=$?; echo "pytest exit $PRC"; grep -h "grch37\|passed\|failed\|skipped\|real" $O/pytest.log | tail -12
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
[ $PRC -ne 0 ] && { tail -40 $O/pytest.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
PMC_STEPS=4 bash $R/tools/pmc_traffic.sh r6_c5 --config C5 > $O/pmc_c5.log 2>&1; tail -3 $O/pmc_c5.log; cp $R/gpurun_out/r6_c5_pmc.json $R/profiles/r6_c5_pmc.json 2>/dev/null
[ $(left) -gt 500 ] && { timeout 2400 python3 $R/bench.py --config C5 --steps 20 --warmup 5 --no-extras > $O/r6_bench_line_c5.json 2> $O/c5.err; python3 $R/tools/ab_show.py c5 < $O/r6_bench_line_c5.json; }
[ $(left) -gt 200 ] && { timeout 1200 python3 $R/bench.py --config C2 --steps 20 --warmup 5 --no-extras > $O/r6_bench_line_c2.json 2> $O/c2.err; python3 $R/tools/ab_show.py c2 < $O/r6_bench_line_c2.json; }
ls $O
