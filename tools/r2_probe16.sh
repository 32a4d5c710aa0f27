#!/bin/bash
# r2 probe 16: config C5 shape on one GPU: 150 bp reads, -n 5 (with the default -o 1 -e 6 -l 32 -k 2) on the GRCh37-scale index.
set -u
mkdir -p gpurun_out/r2p16
( time timeout 2400 python bench.py --read-len 150 --ndiff 5 --pool 10000000 --reads 2500000 --steps 4 --warmup 1 ) > gpurun_out/r2p16/c5.log 2>&1
grep '^{"metric"' gpurun_out/r2p16/c5.log > gpurun_out/r2p16/r2_c5_shape_line.json
tail -4 gpurun_out/r2p16/c5.log | cut -c1-1500
