#!/bin/bash
# r2 probe 18: admission threshold that follows the mean chunk need of the block's finished reads: parity tests, then the C5-shape run of probe 17.
set -u
mkdir -p gpurun_out/r2p18
timeout 900 python -m pytest tests/test_gpu_edge_parity.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
( time BWB_DEBUG=1 timeout 600 python bench.py --read-len 150 --ndiff 5 --pool 10000000 --reads 1000000 --steps 2 --warmup 0 --no-extras ) > gpurun_out/r2p18/c5_dbg.log 2>&1
grep -E "kl_search|iterations|real" gpurun_out/r2p18/c5_dbg.log | cut -c1-330 | tail -12
grep '^{"metric"' gpurun_out/r2p18/c5_dbg.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('value', j['value'], 'rerun', j['rerun_reads'], 'lanes', j['roofline']['lanes_busy_of_64'])"
