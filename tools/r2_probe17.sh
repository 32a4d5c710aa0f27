#!/bin/bash
# r2 probe 17: why does the C5 shape (150 bp, -n 5) crawl with full batches?  1 M-read steps with BWB_DEBUG=1 (launch log: pool use, re-run classes).
set -u
mkdir -p gpurun_out/r2p17
( time BWB_DEBUG=1 timeout 700 python bench.py --read-len 150 --ndiff 5 --pool 10000000 --reads 1000000 --steps 2 --warmup 0 --no-extras ) > gpurun_out/r2p17/c5_dbg.log 2>&1
grep -E "kl_search|kl_calc_d|iterations|metric|real" gpurun_out/r2p17/c5_dbg.log | cut -c1-330 | tail -30
