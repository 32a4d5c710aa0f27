#!/usr/bin/env python3
"""Developer tool: quick perf/parity probe on a synthetic genome (not part of the product path).
usage: quick_perf.py <n_fwd_chars> <n_reads> <n_diff> [check_reads]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bwbble_amd as bw

n_fwd, n_reads, ndiff = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
check = int(sys.argv[4]) if len(sys.argv) > 4 else 0
work = os.environ.get("BWB_WORK", "/tmp/bwb_quick"); os.makedirs(work, exist_ok=True)
fa, fq = f"{work}/g{n_fwd}.fa", f"{work}/g{n_fwd}_{n_reads}.fq"
bw.build()
if not os.path.exists(fa + ".bwt"):
    subprocess.run([bw.SYNTH_BIN, "genome", fa, str(n_fwd), "4", str(max(4, n_fwd // 2400)), "3"], check=True)
    t = time.time()
    builder = os.environ.get("BWB_INDEXER", os.path.join(ROOT, "oracle", "_ref", "bwbble"))
    subprocess.run([builder, "index", fa], check=True, stdout=subprocess.DEVNULL)
    print(f"index built in {time.time()-t:.1f}s with {builder}")
if not os.path.exists(fq):
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(n_reads), "100", "5", "1.0", "0.1", "0.0"], check=True)
t = time.time(); ctx = bw.Context(fa + ".bwt"); print(f"ctx_create {time.time()-t:.2f}s  length={ctx.bwt.length}")
ms, cs = ctx.rank_bench(1 << 24, iters=3)
print(f"rank_bench: {ms:.3f} ms / 16.7M visits -> {(1<<24)/ms/1e6:.2f} Gvisit/s, {(1<<24)*128/ms/1e6:.1f} GB/s (128B buckets), {(1<<24)*192/ms/1e6:.1f} GB/s algorithmic(192B)")
seqs, lens = bw.encode_reads(bw.read_fastq(fq))
for flags in [["-n", d] for d in ndiff.split(",")]:
    p = bw.params(flags)
    for rep in range(2):
        t = time.time(); ctx.upload(p, seqs, lens); tu = time.time() - t
        t = time.time(); ctx.run(); tr = time.time() - t
        t = time.time(); off, alns = ctx.result(); td = time.time() - t
    st = ctx.stats()
    vis = st.visits_single + st.visits_alphabet
    print(f"{flags}: reads={len(lens)} run={tr*1e3:.1f}ms (calc_d {st.ms_calc_d:.1f} search {st.ms_search:.1f} launches {st.launches_calc_d}/{st.launches_search}) "
          f"upload {tu*1e3:.1f} result {td*1e3:.1f} -> {len(lens)/tr:.0f} reads/s; visits={vis} ({vis/len(lens):.0f}/read) "
          f"alg GB/s={vis*192/(st.ms_calc_d+st.ms_search)/1e6:.1f} pops={st.heap_pops} pushes={st.heap_pushes} alns={st.n_alignments} overflow={st.n_overflow_reads}")
    if check:
        import oracle_lib
        orc = oracle_lib.load(); idx = orc.load_index(fa + ".bwt")
        op = orc.params(flags + ["-t", str(os.cpu_count())])
        data, ost, sec = orc.align_encoded(idx, seqs[:check], lens[:check], op)
        ok = bw.aln_bytes(off[:check + 1], alns[:int(off[check])]) == data
        print(f"   oracle check on first {check} reads: {'IDENTICAL' if ok else 'MISMATCH'}; oracle {check/sec:.0f} reads/s on {os.cpu_count()} threads")
