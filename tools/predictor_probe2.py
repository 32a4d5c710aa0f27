#!/usr/bin/env python3
"""Developer tool (round 6): can the heavy reads of a batch be named BEFORE the search from what calculate_d computes anyway?  The draining
launch at the end of a stream is one `max_entries` read that started late; started first in their batch such reads would
be done - or nearly - when the stream ends.  What matters is RECALL of the extreme reads: one missed read keeps the drain as long as it is.
Features: the exact-match widths calculate_d sees on its way down the read (bwb_hip_calc_d: num_diff and sa_intv_width per position) - reads out
of high-copy repeats keep wide intervals deep into the read, unique reads are down to one row after ~16 bases.
usage: BWB_DEBUG_ITERS=1 predictor_probe2.py <genome.fa> <reads.fq> <n_reads> [align flags ...]   (files as bench.py leaves them)"""
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
p = bw.params(flags)
ctx.align(p, seqs, lens)
L = bw.lib()
it = np.zeros(len(lens), dtype=np.uint32)
L.bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
assert L.bwb_hip_debug_iters(ctx._h, it.ctypes.data) == 0
it = it.astype(np.float64)
D, Ds = ctx.calc_d(p, seqs, lens)   # (n, len+1, 2): num_diff, width (the 32-bit wrap of the reference's int)
w = D[:, :, 1].astype(np.int64) & 0xFFFFFFFF
z = D[:, :, 0]
ws = Ds[:, :, 1].astype(np.int64) & 0xFFFFFFFF
ln = int(lens.max())
print(f"reads {len(it)}  search cost: mean {it.mean():.0f} median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} p99.9 {np.percentile(it, 99.9):.0f} max {it.max():.0f}")
lw = np.log2(np.maximum(w[:, :ln], 1))
feats = {
    "log2 width at step 20": lw[:, 20], "log2 width at step 24": lw[:, 24], "log2 width at step 30": lw[:, 30], "log2 width at step 40": lw[:, 40],
    "sum of log2 widths, steps 16..99": lw[:, 16:ln].sum(axis=1), "sum of log2 widths, steps 24..99": lw[:, 24:ln].sum(axis=1),
    "max log2 width, steps 24..99": lw[:, 24:ln].max(axis=1), "restarts z at the read's end (negated)": -z[:, ln - 1].astype(np.float64),
    "log2 seed width at its end": np.log2(np.maximum(ws[:, p.seed_length - 1], 1)),
    "min over steps 16..99 of log2 width (how narrow it ever gets)": lw[:, 16:ln].min(axis=1),
}
order_it = np.argsort(-it)
for name_cut, cut in (("the heaviest 0.02 %% (>= %.0f iterations)", 0.0002), ("the heaviest 0.1 %% (>= %.0f)", 0.001), ("the heaviest 1 %% (>= %.0f)", 0.01)):
    k = max(1, int(len(it) * cut))
    heavy = set(order_it[:k].tolist())
    print("\n== recall of " + (name_cut % it[order_it[k - 1]]) + f": {k} reads")
    for fname, f in feats.items():
        o = np.argsort(-f, kind="stable")
        line = f"   {fname:62s}"
        for top in (0.01, 0.05, 0.10, 0.20, 0.40):
            kk = int(len(it) * top)
            line += f"  top {top:4.0%}: {len(heavy & set(o[:kk].tolist())) / len(heavy):6.1%}"
        # the worst rank of an extreme read under this feature: how large the "first" bin must be for full recall
        rank = np.empty(len(f), dtype=np.int64); rank[o] = np.arange(len(f))
        line += f"  | full recall needs the top {(rank[list(heavy)].max() + 1) / len(f):6.1%}"
        print(line)
# what the extreme reads look like
print("\nthe 12 heaviest reads: iterations | z at the end | log2 widths at steps 12, 16, 20, 24, 30, 40, 60, 99 | seed end")
for r in order_it[:12]:
    print(f"   {int(it[r]):9d} | {int(z[r, ln - 1]):2d} | " + " ".join(f"{lw[r, s]:5.1f}" for s in (12, 16, 20, 24, 30, 40, 60, min(99, ln - 1))) + f" | {np.log2(max(ws[r, p.seed_length - 1], 1)):5.1f}")
print("12 median reads:")
med = np.argsort(np.abs(it - np.median(it)))[:12]
for r in med:
    print(f"   {int(it[r]):9d} | {int(z[r, ln - 1]):2d} | " + " ".join(f"{lw[r, s]:5.1f}" for s in (12, 16, 20, 24, 30, 40, 60, min(99, ln - 1))) + f" | {np.log2(max(ws[r, p.seed_length - 1], 1)):5.1f}")
