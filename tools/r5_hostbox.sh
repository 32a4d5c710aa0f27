#!/bin/bash
# Round 5: the two host stages of `align` (FASTQ reader, chunk serialiser) on the GPU box's 256 cores, by number of OpenMP threads
# (bench.py's host_pipeline keys came out at 8.8 M / 8.1 M per second there and at 17 M / 76 M on the 8-core development container).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r5host; mkdir -p $O
W=/tmp/hb; mkdir -p $W; cd $W
B=$R/bwbble_amd/bin
nproc; cat /sys/kernel/mm/transparent_hugepage/enabled; lscpu | grep -E "NUMA node|Model name|Socket" | head -6
$B/bwb_synth genome g.fa 20000000 1 8000 21 > /dev/null
$B/bwbble index g.fa > /dev/null
( time $B/bwb_synth reads g.fa r.fq 10000000 100 1000 1.0 0.1 0.0 ) 2>&1 | grep real
( time $B/bwbble align -n 0 g.fa r.fq r.aln ) > $O/align.txt 2>&1; grep -E "wall|real" $O/align.txt
ls -la r.fq r.aln
for t in 256 128 64 32 16 8; do
  for rep in 1 2; do echo -n "OMP_NUM_THREADS=$t  "; OMP_NUM_THREADS=$t $B/bwbble hostbench r.fq r.aln; done
done 2>&1 | tee $O/sweep.txt
for ft in 4 16 32 64; do echo -n "BWB_FQ_THREADS=$ft OMP=32  "; BWB_FQ_THREADS=$ft OMP_NUM_THREADS=32 $B/bwbble hostbench r.fq; done 2>&1 | tee -a $O/sweep.txt
echo -n "OMP_PROC_BIND=close OMP=32  "; OMP_PROC_BIND=close OMP_NUM_THREADS=32 $B/bwbble hostbench r.fq r.aln | tee -a $O/sweep.txt
echo -n "GOMP_SPINCOUNT=0 OMP=256  "; GOMP_SPINCOUNT=0 OMP_NUM_THREADS=256 $B/bwbble hostbench r.fq r.aln | tee -a $O/sweep.txt
