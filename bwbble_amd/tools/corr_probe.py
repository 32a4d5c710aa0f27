#!/usr/bin/env python3
"""Developer tool: does k_calc_d's per-read work predict k_search's per-read iterations? Simulates LPT makespans."""
import os, sys, ctypes as C, heapq
import numpy as np
os.environ["BWB_DEBUG_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
fa, fq, nd = sys.argv[1], sys.argv[2], sys.argv[3]
ctx = bw.Context(fa + ".bwt")
seqs, lens = bw.encode_reads(bw.read_fastq(fq))
p = bw.params(["-n", nd])
L = bw.lib(); L.bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]; L.bwb_hip_debug_calcd_work.argtypes = [C.c_void_p, C.c_void_p]
ctx.upload(p, seqs, lens); ctx.run()
it = np.zeros(len(lens), dtype=np.uint32); bw._chk(L.bwb_hip_debug_iters(ctx._h, it.ctypes.data))
cw = np.zeros(len(lens), dtype=np.uint32); bw._chk(L.bwb_hip_debug_calcd_work(ctx._h, cw.ctypes.data))
from scipy.stats import spearmanr
print("reads", len(lens), "iters mean", it.mean(), "max", it.max(), " calc_d work mean", cw.mean(), "max", cw.max())
print("spearman(calc_d work, search iters) =", spearmanr(cw, it).correlation)
top = np.argsort(it)[::-1][:1000]
rank_by_cw = np.argsort(np.argsort(-cw.astype(np.int64)))
print("of the 1000 heaviest reads, median rank by calc_d work:", int(np.median(rank_by_cw[top])), "of", len(lens), "; in top 5% by calc_d work:", float((rank_by_cw[top] < len(lens)*0.05).mean()))
def makespan(order, lanes):
    h = [0]*lanes; heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + int(it[i]))
    return max(h)
for lanes in (65536, 131072):
    print("lanes", lanes, "ideal", int(it.sum()/lanes), "input order", makespan(range(len(lens)), lanes), "by calc_d work desc", makespan(np.argsort(-cw.astype(np.int64), kind="stable"), lanes), "oracle LPT", makespan(np.argsort(-it.astype(np.int64)), lanes))
