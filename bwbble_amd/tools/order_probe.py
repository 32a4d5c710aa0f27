#!/usr/bin/env python3
"""Developer tool: would starting the reads with the largest D lower bound first shorten the drain phase?
Per-read iterations vs z = D[len-1].num_diff (and the seed bound); simulated list-scheduling makespans.
usage: order_probe.py <n_fwd_chars> <n_reads> <n_diff>"""
import os, sys, ctypes as C, heapq
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
D, Ds = ctx.calc_d(p, seqs, lens)
z = D[np.arange(len(lens)), lens.astype(np.int64) - 1, 0]
zs = Ds[:, p.seed_length - 1, 0]
ctx.upload(p, seqs, lens); ctx.run()
bw.lib().bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
it = np.zeros(len(lens), dtype=np.uint32); bw._chk(bw.lib().bwb_hip_debug_iters(ctx._h, it.ctypes.data))
print("reads", len(lens), "iterations mean", it.mean(), "max", it.max())
for v in range(0, 6):
    m = z == v
    if m.any(): print(f"z={v}: {m.mean()*100:.2f} % of reads, mean it {it[m].mean():.0f}, p99 {np.percentile(it[m],99):.0f}, max {it[m].max()}, share of all iterations {it[m].sum()/it.sum()*100:.1f} %")
top = np.argsort(it)[::-1][:len(lens)//100]
print("z of the 1 % heaviest reads:", np.bincount(z[top].clip(0, 8)))
print("seed z of the 1 % heaviest reads:", np.bincount(zs[top].clip(0, 8)))
feat = {}
for k in (15, 20, 25, 30, 40, 60):
    feat[f"width after {k} bases"] = D[:, k - 1, 1].astype(np.int64)
    feat[f"seed width after {k} bases"] = Ds[:, min(k, p.seed_length) - 1, 1].astype(np.int64)
feat["sum of widths k>=20"] = D[:, 19:, 1].astype(np.int64).clip(0, 1 << 20).sum(axis=1)
feat["min width k in 20..len-1 where z unchanged"] = np.where(D[:, 19:, 0] == D[:, 19:20, 0], D[:, 19:, 1], 1 << 30).min(axis=1).astype(np.int64)
from scipy.stats import spearmanr
for k, f in feat.items():
    r = np.argsort(np.argsort(-f, kind="stable"))
    print(f"{k}: spearman {spearmanr(f, it).correlation:.3f}; of the 1 % heaviest, in the top 5 % by it: {(r[top] < len(lens) * 0.05).mean():.2f}, top 20 %: {(r[top] < len(lens) * 0.2).mean():.2f}")
def makespan(order, lanes):
    h = [0] * lanes; heapq.heapify(h)
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + int(it[i]))
    return max(h)
lanes = 131072
keys = {"input order": np.arange(len(lens)), "z desc": np.argsort(-z, kind="stable"), "(z, seed z) desc": np.lexsort((-zs, -z)),
        "width after 25 desc": np.argsort(-feat["width after 25 bases"], kind="stable"), "sum widths desc": np.argsort(-feat["sum of widths k>=20"], kind="stable"),
        "oracle (iterations desc)": np.argsort(-it.astype(np.int64), kind="stable")}
print("ideal", int(it.sum() / lanes))
for k, o in keys.items():
    print(f"{k}: makespan {makespan(o, lanes)}")
