#!/bin/bash
# r2 probe 11: `bwbble aln2sam` at GRCh37 scale: SA lookups of 9.76 M mapped reads on the GPU (k_locate), SAM text on the host;
# first 5000 records against the reference's own aln2sam on the same reads.
set -u
mkdir -p gpurun_out/r2p11
timeout 1500 python bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>&1   # builds /tmp/bwb_bench (genome, index, reads)
W=/tmp/bwb_bench; FA=$W/genome_3100000000.fa; FQ=$W/reads_3100000000_10000000_100_r0.fq
( time bwbble_amd/bin/bwbble align -n 3 $FA $FQ $W/cli.aln ) > gpurun_out/r2p11/cli_align.log 2>&1
( time bwbble_amd/bin/bwbble aln2sam -n 3 $FA $FQ $W/cli.aln $W/cli.sam ) > gpurun_out/r2p11/cli_aln2sam.log 2>&1
grep -E "real|Processed" gpurun_out/r2p11/cli_aln2sam.log | tail -3
head -20000 $FQ > $W/head5000.fq
oracle/_ref/bwbble align -n 3 -t 256 $FA $W/head5000.fq $W/ref5000.aln > /dev/null
( time oracle/_ref/bwbble aln2sam -n 3 $FA $W/head5000.fq $W/ref5000.aln $W/ref5000.sam ) 2>&1 | grep real
python3 - <<'PY'
W = "/tmp/bwb_bench"
ref = open(W + "/ref5000.sam").read().splitlines()
got = open(W + "/cli.sam").read().splitlines()
nh = sum(1 for l in ref if l.startswith("@"))
print("reference SAM lines", len(ref), "header lines", nh, "CLI SAM lines", len(got))
print("header identical:", ref[:nh] == got[:nh], " first 5000 records identical:", ref[nh:] == got[nh:nh + 5000])
PY
