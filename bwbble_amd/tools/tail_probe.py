#!/usr/bin/env python3
"""Developer tool: how much of kl_search's time is the serial chain of the heaviest reads?  Times the batch with its K
heaviest reads (by loop iterations) removed.  usage: tail_probe.py <n_fwd_chars> <n_reads> <n_diff>"""
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
tot = it.sum()
print("iterations: total", int(tot), "max", int(it.max()), "p50", int(np.median(it)), "p99", int(np.percentile(it, 99)), "p99.9", int(np.percentile(it, 99.9)))
for K in (0, 10, 100, 1000, 10000, 100000):
    keep = np.sort(order[K:])
    ctx.upload(p, seqs[keep], lens[keep]); ctx.run(); ctx.run()
    st = ctx.stats()
    print(f"without the {K} heaviest: search {st.ms_search:.1f} ms, iterations left {int(it[keep].sum())} ({100.0*it[keep].sum()/tot:.1f} %), max {int(it[keep].max())}", flush=True)
