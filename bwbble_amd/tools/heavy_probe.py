#!/usr/bin/env python3
"""Developer tool: which reads are heavy, and how fast is the serial chain of a heavy read when it runs alone?"""
import os, sys, time, ctypes as C
import numpy as np
os.environ["BWB_DEBUG_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_WORK", "/tmp/bwb_quick")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/g{n_fwd}.fa", f"{work}/g{n_fwd}_{n_reads}.fq"
ctx = bw.Context(fa + ".bwt")
seqs, lens = bw.encode_reads(bw.read_fastq(fq))
p = bw.params(["-n", nd])
bw.lib().bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
def run(sel, label):
    ctx.upload(p, seqs[sel], lens[sel]); ctx.run(); ctx.run()
    st = ctx.stats()
    it = np.zeros(len(sel), dtype=np.uint32)
    bw._chk(bw.lib().bwb_hip_debug_iters(ctx._h, it.ctypes.data))
    print(f"{label}: reads {len(sel)} search {st.ms_search:.2f} ms  iters sum {int(it.sum())} max {int(it.max())}  -> {st.ms_search*1e3/max(1,int(it.max())):.2f} us per iteration of the longest read; {int(it.sum())/st.ms_search/1e6:.2f} G iter/s")
    return it
it = run(np.arange(len(lens)), "all")
order = np.argsort(it)[::-1]
print("top iteration counts:", it[order[:8]], " median", int(np.median(it)), "mean", int(it.mean()))
run(order[:1], "heaviest alone")
run(order[:64], "64 heaviest")
run(order[:4096], "4096 heaviest")
med = np.argsort(np.abs(it.astype(np.int64) - int(np.median(it))))[:65536]
run(med, "65536 median-like reads")
run(np.tile(order[len(order)//2:len(order)//2+1], 65536), "65536 copies of one median read")
