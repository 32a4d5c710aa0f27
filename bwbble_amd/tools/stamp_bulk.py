#!/usr/bin/env python3
"""Developer tool: per-segment cycle breakdown of kl_search (diagnostic -DBWB_STAMPS build) over a whole batch, then for
its heaviest read alone and for its 64 heaviest reads.
usage: stamp_bulk.py <n_fwd_chars> <n_reads> <n_diff>   (uses the files bench.py left in /tmp/bwb_bench)"""
import os, sys, ctypes as C
import numpy as np
os.environ["BWB_DEBUG_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
bw.LIB_PATH = os.path.join(ROOT, "bwbble_amd", "tools_exp", "libbwbble_hip_stamps.so")
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{n_reads}_100_r0.fq"
ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
seqs, lens = bw.load_fastq_codes(fq)
p = bw.params(["-n", nd])
ctx.upload(p, seqs, lens); ctx.run()
os.environ["BWB_DEBUG"] = "1"
print("== whole batch", flush=True)
ctx.run()
bw.lib().bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
it = np.zeros(len(lens), dtype=np.uint32); bw._chk(bw.lib().bwb_hip_debug_iters(ctx._h, it.ctypes.data))
order = np.argsort(it)[::-1]
for sel, label in ((order[:1], "heaviest alone"), (order[:64], "64 heaviest")):
    print("==", label, "iters", int(it[sel].sum()), "max", int(it[sel].max()), flush=True)
    ctx.upload(p, seqs[sel], lens[sel]); ctx.run()
