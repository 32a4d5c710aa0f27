#!/bin/bash
# Developer tool: GRCh37-scale (config C3) run of bench.py on the GPU box, guarded by a host-memory check.
# usage: c3_probe.sh <genome_mb> <reads> [ndiff]
set -u
MB=${1:-3100}; READS=${2:-1000000}; ND=${3:-3}
free -g | head -2; nproc; df -h /tmp | tail -1
avail=$(free -g | awk '/^Mem:/{print $7}')
need=$(( MB * 40 / 1000 + 20 ))
echo "host memory available ${avail} GB, estimated need ${need} GB"
if [ "$avail" -lt "$need" ]; then echo "not enough host memory, skipping"; exit 0; fi
BWB_DEBUG=1 timeout 2400 python bench.py --genome-mb $MB --reads $READS --ndiff $ND --steps 1 --warmup 0 --cpu-sample 500 2>&1 | grep -vE "^\s*$" | tail -30 | cut -c1-4000
