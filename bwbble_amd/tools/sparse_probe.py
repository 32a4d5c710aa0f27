#!/usr/bin/env python3
"""Developer tool: loop-iteration time of a wave holding only the k heaviest reads of a batch (k = 1..64).
usage: sparse_probe.py <n_fwd_chars> <n_reads> <n_diff>"""
import os, sys, ctypes as C
import numpy as np
os.environ["BWB_DEBUG_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{n_reads}_100_r0.fq"
ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
seqs, lens = bw.load_fastq_codes(fq)
p = bw.params(["-n", nd])
ctx.upload(p, seqs, lens); ctx.run()
bw.lib().bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
it = np.zeros(len(lens), dtype=np.uint32); bw._chk(bw.lib().bwb_hip_debug_iters(ctx._h, it.ctypes.data))
order = np.argsort(it)[::-1]
for k in (1, 2, 4, 8, 12, 16, 24, 32, 64):
    sel = order[:k]
    ctx.upload(p, seqs[sel], lens[sel]); ctx.run(); ctx.run()
    st = ctx.stats()
    mx = int(it[sel].max())
    print(f"{k:3d} heaviest reads in one wave: {st.ms_search:8.1f} ms for {mx} wave iterations = {st.ms_search * 1e3 / mx:.2f} us per iteration; mean active lanes {it[sel].sum() / mx:.1f}", flush=True)
