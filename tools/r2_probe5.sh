#!/bin/bash
# r2 probe 5: cache policy of the bucket gather (aux bits) at 400 M-char scale (884 M rows: HBM regime), 1 M-read steps.
set -u
mkdir -p gpurun_out/r2p5
for lib in "" bwbble_amd/tools_exp/libbwbble_hip_aux2.so bwbble_amd/tools_exp/libbwbble_hip_aux3.so bwbble_amd/tools_exp/libbwbble_hip_aux18.so; do
  echo "== lib ${lib:-product}"
  BWB_LIB=${lib:+$PWD/$lib} timeout 900 python bench.py --genome-mb 400 --pool 4000000 --reads 1000000 --steps 6 --warmup 1 --no-extras 2>&1 | grep '^{"metric"' | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['roofline']['kernels']
print('value', j['value'], 'ms/step', j['ms_per_step'], 'search ms/launch', k['kl_search']['ms_per_launch'], 'frac', k['kl_search']['frac'], 'lanes', j['roofline']['lanes_busy_of_64'], 'calc_d ms', k['kl_calc_d']['ms_per_launch'], 'rerun', j['rerun_reads'])"
done 2>&1 | tee gpurun_out/r2p5/aux.log
