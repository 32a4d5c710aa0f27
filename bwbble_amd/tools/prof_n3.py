#!/usr/bin/env python3
"""Developer tool: one -n 3 batch on the quick_perf genome (for rocprofv3 runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_WORK", "/tmp/bwb_quick")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/g{n_fwd}.fa", f"{work}/g{n_fwd}_{n_reads}.fq"
ctx = bw.Context(fa + ".bwt")
seqs, lens = bw.encode_reads(bw.read_fastq(fq))
ctx.upload(bw.params(["-n", nd]), seqs, lens)
ctx.run()
st = ctx.stats()
print("ms", st.ms_calc_d, st.ms_search, "visits", st.visits_single + st.visits_alphabet)
