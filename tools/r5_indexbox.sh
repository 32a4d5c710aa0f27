#!/bin/bash
# Round 5: `bwbble index` (host/index.c) on the GPU box's 256 hardware threads by OpenMP team size - does the index builder suffer from the
# default team like the host stages of `align` did (profiles/r5_host_stage_threads.txt)?  400 M forward characters, 4 records.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r5index; mkdir -p $O
W=/tmp/ib; mkdir -p $W; cd $W
B=$R/bwbble_amd/bin
$B/bwb_synth genome g.fa 400000000 4 160000 21 > /dev/null; ls -la g.fa
for t in 256 128 64 32; do
  rm -f g.fa.bwt g.fa.ann g.fa.ref
  echo -n "OMP_NUM_THREADS=$t  "; TIMEFORMAT="wall %R s  user %U s  sys %S s"; time OMP_NUM_THREADS=$t $B/bwbble index g.fa > /dev/null
  md5sum g.fa.bwt | cut -c1-12
done 2>&1 | tee $O/sweep.txt
rm -f g.fa.bwt; ( echo -n "GOMP_SPINCOUNT=0 (256)  "; TIMEFORMAT="wall %R s  user %U s  sys %S s"; time GOMP_SPINCOUNT=0 $B/bwbble index g.fa > /dev/null ) 2>&1 | tee -a $O/sweep.txt
