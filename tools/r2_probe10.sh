#!/bin/bash
# r2 probe 10: randomised parity sweep (plain and time-sliced), then the CLI end to end on the GRCh37-scale workload:
# `bwbble align -n 3` on 10 M reads (index load, FASTQ parse, 5 streamed chunks, .aln written) and the sample check against the reference.
set -u
mkdir -p gpurun_out/r2p10
( time timeout 900 python tools/fuzz_parity.py 30 11 ) > gpurun_out/r2p10/fuzz_plain.log 2>&1; tail -2 gpurun_out/r2p10/fuzz_plain.log
( time BWB_SLICE_ITERS=97 timeout 900 python tools/fuzz_parity.py 20 12 ) > gpurun_out/r2p10/fuzz_sliced.log 2>&1; tail -2 gpurun_out/r2p10/fuzz_sliced.log
timeout 1500 python bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>&1   # builds /tmp/bwb_bench (genome, index, reads)
W=/tmp/bwb_bench; FA=$W/genome_3100000000.fa; FQ=$W/reads_3100000000_10000000_100_r0.fq
( time bwbble_amd/bin/bwbble align -n 3 $FA $FQ $W/cli.aln ) > gpurun_out/r2p10/cli_align.log 2>&1
grep -E "GPUs:|Total|real" gpurun_out/r2p10/cli_align.log
head -20000 $FQ > $W/head5000.fq
( time oracle/_ref/bwbble align -n 3 -t 256 $FA $W/head5000.fq $W/ref5000.aln ) 2>&1 | grep -E "real"
python3 - <<'PY'
import sys
sys.path.insert(0, "tests")
import oracle_lib
W = "/tmp/bwb_bench"
ref = open(W + "/ref5000.aln", "rb").read()
got = open(W + "/cli.aln", "rb").read()
print("CLI .aln bytes", len(got), "first 5000 reads identical to the reference's:", got[:len(ref)] == ref)
PY
