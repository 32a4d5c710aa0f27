#!/bin/bash
# Developer tool: AddressSanitizer + UBSan over the CPU-side code (the oracle and the host tools' CPU-only commands).
# GPU sanitizers are not available on the pool, so this is the sanitizer coverage there is.  Run from the repo root.
set -eu
R=$(pwd); T=$(mktemp -d); G=$R/tests/golden
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -DBWB_ORACLE_MAIN -o $T/oracle_asan $R/oracle/bwb_oracle.c -lm
gcc -O1 -g -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -I$R/include -o $T/bwbble_asan $R/bwbble_amd/host/*.c \
    -L$R/bwbble_amd -lbwbble_hip -Wl,-rpath,$R/bwbble_amd -lm -lpthread
export ASAN_OPTIONS=detect_leaks=0
for cfg in "-n 3" "-n 4 -o 2 -e 3 -l 20 -k 1" "-S -n 2" "-P -n 2"; do $T/oracle_asan $cfg $G/toy.fa $G/ragged.fq $T/o.aln > /dev/null; done
cmp $T/o.aln $G/ragged_p2.aln
cp $G/toy.fa $T/ && (cd $T && ./bwbble_asan index toy.fa > /dev/null && ./bwbble_asan fasta2ref toy.fa > /dev/null)
cmp $T/toy.fa.bwt $G/toy.fa.bwt && cmp $T/toy.fa.ann $G/toy.fa.ann
# the prefix-doubling phase of the suffix sort (long exact repeats) and the external-SA ingest
python3 -c "import sys; sys.path.insert(0, '$R'); sys.path.insert(0, '$R/tests'); import test_host_tools as t; t._repeat_rich_fasta('$T/rep.fa')"
(cd $T && BWB_DUMP_SA=$T/rep.sa5 ./bwbble_asan index rep.fa > /dev/null && cp rep.fa.bwt a.bwt && ./bwbble_asan index -e rep.sa5 rep.fa > /dev/null && cmp a.bwt rep.fa.bwt)
# round 5's host code: the multi-unit .bwt loader, the parallel FASTQ scanner with parts of a few bytes, the chunk serialiser of `align`
(cd $T && for u in 1 7 3129; do BWB_LOAD_THREADS=5 BWB_LOAD_UNIT=$u ./bwbble_asan bwtcat $G/toy.fa.bwt c.bwt > /dev/null && cmp c.bwt $G/toy.fa.bwt; done)
(cd $T && ./bwbble_asan dumpreads $G/sim_chr21_N100.fastq whole.tsv > /dev/null && cut -f2 whole.tsv > whole.txt
 for rg in 1 64 300 4096; do BWB_FQ_REGION=$rg BWB_FQ_THREADS=7 ./bwbble_asan dumpreads $G/sim_chr21_N100.fastq parts.txt 13 > /dev/null && cmp parts.txt whole.txt; done)
(cd $T && for a in toy_n4gap.aln gapo_o6.aln; do ./bwbble_asan alncat $G/$a r.aln > /dev/null && ./bwbble_asan alncat $G/$a c.aln buf 7 > /dev/null && cmp c.aln r.aln; done)
# the pthread loader under ThreadSanitizer (blocks_ready against the unit flags).  The FASTQ scanner's threads are an OpenMP team: libgomp is not
# instrumented, so TSan does not see the team's join barrier and reports the serial stitch after it as a race with the team - not run here
gcc -O1 -g -std=gnu11 -fsanitize=thread -fno-omit-frame-pointer -fopenmp -I$R/include -o $T/bwbble_tsan $R/bwbble_amd/host/*.c \
    -L$R/bwbble_amd -lbwbble_hip -Wl,-rpath,$R/bwbble_amd -lm -lpthread
(cd $T && export TSAN_OPTIONS="halt_on_error=1 ignore_noninstrumented_modules=1"
 for u in 1 7; do BWB_LOAD_THREADS=5 BWB_LOAD_UNIT=$u ./bwbble_tsan bwtcat $G/toy.fa.bwt c.bwt > /dev/null && cmp c.bwt $G/toy.fa.bwt; done)
echo "sanitizer run clean"
rm -rf $T
