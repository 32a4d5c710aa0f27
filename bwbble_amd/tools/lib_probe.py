#!/usr/bin/env python3
"""Developer tool: times one -n <d> batch (bench.py's files in /tmp/bwb_bench) with an alternative build of the library.
usage: lib_probe.py <lib.so> <n_fwd_chars> <n_reads> <n_diff>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
bw.LIB_PATH = os.path.abspath(sys.argv[1])
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, n_reads, nd = sys.argv[2], int(sys.argv[3]), sys.argv[4]
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{n_reads}_100_r0.fq"
ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
seqs, lens = bw.load_fastq_codes(fq)
ctx.upload(bw.params(["-n", nd]), seqs, lens)
ctx.run(); ctx.run()
st = ctx.stats()
print(os.path.basename(sys.argv[1]), "calc_d ms", round(st.ms_calc_d, 2), "search ms", round(st.ms_search, 2))
