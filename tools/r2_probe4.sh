#!/bin/bash
# r2 probe 4: one block per CU (1 wave/SIMD) against two, at chr21 scale (cache) and 400 M-char scale (HBM); new GPU tests.
set -u
mkdir -p gpurun_out/r2p4
( time timeout 900 python -m pytest tests/test_host_tools.py -m gpu -x -q ) > gpurun_out/r2p4/gputests.log 2>&1
tail -5 gpurun_out/r2p4/gputests.log
for mb in 48 400; do
  for bpc in 2 1 3; do
    echo "== genome-mb $mb blocks/CU $bpc"
    BWB_BLOCKS_PER_CU=$bpc timeout 900 python bench.py --genome-mb $mb --pool 4000000 --reads 1000000 --steps 4 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['kernels']
print('value', j['value'], 'ms/step', j['ms_per_step'], 'search ms/launch', k['kl_search']['ms_per_launch'], 'frac', k['kl_search']['frac'], 'lanes', j['roofline']['lanes_busy_of_64'], 'calc_d ms', k['kl_calc_d']['ms_per_launch'], 'rerun', j['rerun_reads'])"
  done
done 2>&1 | tee gpurun_out/r2p4/bpc.log
