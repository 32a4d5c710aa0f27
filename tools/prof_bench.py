#!/usr/bin/env python3
"""Developer tool: one -n <d> batch on the files bench.py left in /tmp/bwb_bench (for rocprofv3 --pmc passes).
usage: prof_bench.py <n_fwd_chars> <n_reads> <n_diff>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{n_reads}_100_r0.fq"
ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
seqs, lens = bw.load_fastq_codes(fq)
ctx.upload(bw.params(["-n", nd]), seqs, lens)
ctx.run()
st = ctx.stats()
print("ms", st.ms_calc_d, st.ms_search, "visits", st.visits_single + st.visits_alphabet)
