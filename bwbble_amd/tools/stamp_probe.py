#!/usr/bin/env python3
"""Developer tool: per-segment cycle breakdown (diagnostic build with s_memtime stamps) for the heaviest read alone."""
import os, sys, ctypes as C
import numpy as np
os.environ["BWB_DEBUG_ITERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
bw.LIB_PATH = os.path.join(ROOT, "bwbble_amd", "tools_exp", "libbwbble_hip_stamps.so")
work = os.environ.get("BWB_WORK", "/tmp/bwb_quick")
n_fwd, n_reads, nd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fa, fq = f"{work}/g{n_fwd}.fa", f"{work}/g{n_fwd}_{n_reads}.fq"
ctx = bw.Context(fa + ".bwt")
seqs, lens = bw.encode_reads(bw.read_fastq(fq))
p = bw.params(["-n", nd])
bw.lib().bwb_hip_debug_iters.argtypes = [C.c_void_p, C.c_void_p]
os.environ.pop("BWB_DEBUG", None)
ctx.upload(p, seqs, lens); ctx.run()
it = np.zeros(len(lens), dtype=np.uint32); bw._chk(bw.lib().bwb_hip_debug_iters(ctx._h, it.ctypes.data))
order = np.argsort(it)[::-1]
os.environ["BWB_DEBUG"] = "1"
for sel, label in ((order[:1], "heaviest alone"), (order[:64], "64 heaviest"), (np.tile(order[len(order)//2:len(order)//2+1], 64), "64 copies of a median read")):
    print("==", label, "iters", int(it[sel].sum()), flush=True)
    ctx.upload(p, seqs[sel], lens[sel]); ctx.run()
