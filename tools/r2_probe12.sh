#!/bin/bash
# r2 probe 12: (a) finer s_memtime stamps of kl_search at C3; (b) `bwbble aln2sam` at C3 with the binary search of the annotation
# record (9 minutes with the reference's linear scan over 1.29 M records).
set -u
mkdir -p gpurun_out/r2p12
BWB_DEBUG=1 BWB_LIB=$PWD/bwbble_amd/tools_exp/libbwbble_hip_stamps.so timeout 1800 python bench.py --steps 3 --warmup 0 --no-extras > gpurun_out/r2p12/stamps.log 2>&1
grep -E "stamps|iterations" gpurun_out/r2p12/stamps.log | cut -c1-900
W=/tmp/bwb_bench; FA=$W/genome_3100000000.fa; FQ=$W/reads_3100000000_10000000_100_r0.fq
( time bwbble_amd/bin/bwbble align -n 3 $FA $FQ $W/cli.aln ) > gpurun_out/r2p12/cli_align.log 2>&1
( time timeout 1200 bwbble_amd/bin/bwbble aln2sam -n 3 $FA $FQ $W/cli.aln $W/cli.sam ) > gpurun_out/r2p12/cli_aln2sam.log 2>&1
grep -E "real|Processed" gpurun_out/r2p12/cli_aln2sam.log | tail -3
head -20000 $FQ > $W/head5000.fq
oracle/_ref/bwbble align -n 3 -t 256 $FA $W/head5000.fq $W/ref5000.aln > /dev/null
oracle/_ref/bwbble aln2sam -n 3 $FA $W/head5000.fq $W/ref5000.aln $W/ref5000.sam > /dev/null
python3 - <<'PY'
W = "/tmp/bwb_bench"
ref = open(W + "/ref5000.sam").read().splitlines()
got = open(W + "/cli.sam").read().splitlines()
nh = sum(1 for l in ref if l.startswith("@"))
print("header identical:", ref[:nh] == got[:nh], " first 5000 records identical:", ref[nh:] == got[nh:nh + 5000], " CLI SAM lines", len(got))
PY
