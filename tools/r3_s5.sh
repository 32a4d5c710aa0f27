#!/bin/bash
# round-3 session 5: eight slots at C3 (driver arguments), launch log of C5, predictor probe, C2
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3s5; mkdir -p $O
( cd $R && time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/gputests.log 2>&1; tail -4 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
bash $R/tools/ab_bench.sh r3s5 "--steps 20 --warmup 5 --no-extras" product20
timeout 900 python3 $R/tools/predictor_probe.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_10000000_100_i0.1_r0.fq 400000 -n 3 2>&1 | grep -v "^\[bwb" | tee $O/predictor_c3.txt
BWB_DEBUG=1 timeout 1500 python3 $R/bench.py --config C5 --steps 9 --warmup 0 --no-extras > $O/c5_debug.json 2> $O/c5_debug.err; grep "^\[bwb\]" $O/c5_debug.err | cut -c1-330 | head -40
bash $R/tools/ab_bench.sh r3s5 "--config C5 --steps 12 --warmup 4 --no-extras" c5_12
timeout 900 python3 $R/tools/predictor_probe.py /tmp/bwb_bench/genome_3100000000.fa /tmp/bwb_bench/reads_3100000000_4000000_150_i0.2_r0.fq 200000 -n 5 -o 1 -e 6 -l 32 -k 2 2>&1 | grep -v "^\[bwb" | tee $O/predictor_c5.txt
bash $R/tools/ab_bench.sh r3s5 "--config C2 --steps 20 --warmup 5 --no-extras" c2_20
