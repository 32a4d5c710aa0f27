#!/usr/bin/env python3
"""Developer tool: do two batches in flight on one GPU (two contexts, two host threads) overlap the drain phase of one
with the bulk of the other?  usage: pipe_probe.py <n_fwd_chars> <n_reads> <n_diff> [steps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{n_reads}_100_r0.fq"
bwt = bw.BwtFile(fa + ".bwt")
seqs, lens = bw.load_fastq_codes(fq)
p = bw.params(["-n", nd])
os.environ.setdefault("BWB_POOL_GB", "60")
ctxs = [bw.Context(bwt) for _ in range(2)]
for c in ctxs:
    c.upload(p, seqs, lens); c.run()
t = time.perf_counter()
for _ in range(steps):
    ctxs[0].run()
seq_t = (time.perf_counter() - t) / steps
def worker(c, n):
    for _ in range(n):
        c.run()
t = time.perf_counter()
th = [threading.Thread(target=worker, args=(c, steps // 2)) for c in ctxs]
[x.start() for x in th]; [x.join() for x in th]
par_t = (time.perf_counter() - t) / steps
print(f"sequential {seq_t*1e3:.1f} ms/step; two in flight {par_t*1e3:.1f} ms/step ({seq_t/par_t:.2f}x)")
