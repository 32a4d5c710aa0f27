#!/bin/bash
# round-3 last GPU minutes (2.6): no torch, no python - the CLI on a chr21-scale index with the product library and with the three
# builds of branch exp-r4 (record unpacked after the rank / side pushes deferred / both): do the .aln files agree, and what do the
# kernels take (BWB_DEBUG launch log)?  -> profiles/r3_r4prep_cli_c2.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x2; mkdir -p $O
S=$R/bwbble_amd/bin/bwb_synth; B=$R/bwbble_amd/bin/bwbble
cd /tmp
$S genome c2.fa 48000000 1 20000 21 && $B index c2.fa > /dev/null && $S reads c2.fa c2.fq 4000000 100 1000 1.0 0.1 0.0
for v in product laterec lateside both; do
  ( [ $v = product ] || export LD_LIBRARY_PATH=$R/xlibs/$v; BWB_DEBUG=1 timeout 40 $B align -n 3 c2.fa c2.fq out_$v.aln > /dev/null 2> $O/log_$v.txt )
  echo "$v md5 $(md5sum < out_$v.aln | cut -c1-12) kl_search ms $(grep 'kl_search class' $O/log_$v.txt | sed 's/.* \([0-9.]*\) ms.*/\1/' | paste -sd+ | bc) launches $(grep -c 'kl_search class' $O/log_$v.txt) kl_calc_d ms $(grep 'kl_calc_d class' $O/log_$v.txt | sed 's/.* \([0-9.]*\) ms.*/\1/' | paste -sd+ | bc)" | tee -a $O/summary.txt
done
