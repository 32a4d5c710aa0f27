#!/usr/bin/env python3
"""Developer tool: how well does calculate_d's work per read (visits kl_calc_d made: what it already stores next to the D records)
predict the read's search cost (pops + exact steps of kl_search)?  Decides whether a batch can be started heaviest-first.
usage: BWB_DEBUG_ITERS=1 predictor_probe.py <genome.fa> <reads.fq> <n_reads> [align flags ...]   (files as bench.py leaves them)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BWB_DEBUG_ITERS", "1")
import bwbble_amd as bw

fa, fq, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
flags = sys.argv[4:] or ["-n", "3"]
seqs, lens = bw.load_fastq_codes(fq, max_reads=n)
ctx = bw.Context(fa + ".bwt")
ctx.align(bw.params(flags), seqs, lens)
L = bw.lib()
it = np.zeros(len(lens), dtype=np.uint32)
wk = np.zeros(len(lens), dtype=np.uint32)
L.bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
L.bwb_hip_debug_calcd_work.argtypes = [C.c_void_p, C.c_void_p]
assert L.bwb_hip_debug_iters(ctx._h, it.ctypes.data) == 0 and L.bwb_hip_debug_calcd_work(ctx._h, wk.ctypes.data) == 0
it = it.astype(np.float64); wk = wk.astype(np.float64)
rk = lambda v: np.argsort(np.argsort(v))
print(f"reads {len(it)}  search cost: mean {it.mean():.0f} median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} p99.9 {np.percentile(it, 99.9):.0f} max {it.max():.0f}")
print(f"calc_d work : mean {wk.mean():.0f} median {np.median(wk):.0f} p99 {np.percentile(wk, 99):.0f} max {wk.max():.0f}")
print(f"pearson {np.corrcoef(it, wk)[0, 1]:.3f}  spearman {np.corrcoef(rk(it), rk(wk))[0, 1]:.3f}")
order = np.argsort(-wk)
tot = it.sum()
for top in (0.01, 0.05, 0.1, 0.25, 0.5):
    k = int(len(it) * top)
    heavy = set(np.argsort(-it)[:max(1, len(it) // 100)])
    print(f"the {top:.0%} of reads with the most calc_d work hold {it[order[:k]].sum() / tot:.1%} of the search cost and {len(heavy & set(order[:k])) / len(heavy):.0%} of the heaviest 1 % of reads")
