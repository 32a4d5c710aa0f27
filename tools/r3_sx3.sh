#!/bin/bash
# round-3, the very last GPU seconds: like tools/r3_sx2.sh (the CLI at chr21 scale, no python), builds of branch exp-r4 that cut
# instructions: private chunk run first in alloc / chunk link fetched one allocation ahead / leaner gather / combinations; the product
# library first and last (run-to-run noise) -> profiles/r3_r4prep_cli_c2.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3; mkdir -p $O
S=$R/bwbble_amd/bin/bwb_synth; B=$R/bwbble_amd/bin/bwbble
cd /tmp
$S genome c2.fa 48000000 1 20000 21 && $B index c2.fa > /dev/null && $S reads c2.fa c2.fq 4000000 100 1000 1.0 0.1 0.0
for v in product priv fnext g2 g2priv all product2; do
  ( case $v in product*) ;; *) export LD_LIBRARY_PATH=$R/xlibs/$v;; esac; BWB_DEBUG=1 timeout 30 $B align -n 3 c2.fa c2.fq out_$v.aln > /dev/null 2> $O/log_$v.txt )
  echo "$v md5 $(md5sum < out_$v.aln | cut -c1-12) $(grep -c 'kl_search class' $O/log_$v.txt) launches: $(grep 'kl_search class\|kl_calc_d class' $O/log_$v.txt | sed 's/.*\(kl_[a-z_]*\) class.* \([0-9.]*\) ms.*/\1 \2/' | tr '\n' ' ')" | tee -a $O/summary.txt
done
