#!/usr/bin/env python3
"""Developer tool: randomised GPU-vs-oracle parity sweep (parameters, read lengths, error rates) on a 3 M-char genome.
usage: fuzz_parity.py [n_configs] [seed]"""
import os, random, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bwbble_amd as bw
import oracle_lib
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bw.build()
d = tempfile.mkdtemp(prefix="bwb_fuzz_")
fa = os.path.join(d, "g.fa")
subprocess.run([bw.SYNTH_BIN, "genome", fa, "3000000", "4", "1500", str(rng.randrange(1000))], check=True)
subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL)
orc = oracle_lib.load(); idx = orc.load_index(fa + ".bwt")
ctx = bw.Context(fa + ".bwt")
bad = 0
for c in range(n_cfg):
    ln = rng.choice([24, 36, 50, 76, 100, 100, 125, 150, 200])
    fq = os.path.join(d, f"r{c}.fq")
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(rng.choice([300, 800, 1500])), str(ln), str(rng.randrange(10000)),
                    str(rng.choice([0.5, 1.0, 2.0, 4.0])), str(rng.choice([0.0, 1.0, 5.0])), str(rng.choice([0.0, 1.0, 10.0]))], check=True)
    seqs, lens = bw.load_fastq_codes(fq)
    if rng.random() < 0.5:  # mix in reads of a second length, interleaved (short reads then inherit D_seed from longer ones before them)
        ln2 = rng.choice([20, 28, 33, 64, 100])
        fq2 = os.path.join(d, f"r{c}b.fq")
        subprocess.run([bw.SYNTH_BIN, "reads", fa, fq2, str(rng.choice([100, 400])), str(ln2), str(rng.randrange(10000)), "2.0", "2.0", "2.0"], check=True)
        s2, l2 = bw.load_fastq_codes(fq2)
        w = max(seqs.shape[1], s2.shape[1])
        allseq = np.full((len(lens) + len(l2), w), 4, dtype=np.uint8)
        allseq[:len(lens), :seqs.shape[1]] = seqs
        allseq[len(lens):, :s2.shape[1]] = s2
        order = np.random.default_rng(c).permutation(len(allseq))
        seqs, lens = allseq[order], np.concatenate([lens, l2])[order]
        ln = f"{ln}+{ln2}"
    flags = ["-n", str(rng.choice([0, 1, 2, 3, 3, 4])), "-o", str(rng.choice([0, 1, 1, 2, 3])), "-e", str(rng.choice([0, 2, 6])),
             "-l", str(rng.choice([0, 16, 32, 32, 60])), "-k", str(rng.choice([0, 1, 2, 3])), "-M", str(rng.choice([1, 3, 3, 5])),
             "-O", str(rng.choice([3, 11, 11])), "-E", str(rng.choice([1, 4, 4])), "-m", str(rng.choice([200, 5000, 3000000]))]
    if rng.random() < 0.2: flags.append("-S")
    if rng.random() < 0.2: flags.append("-P")
    try:
        p = bw.params(flags)
        off, alns = ctx.align(p, seqs, lens)
    except bw.BwbError as e:
        print("config", c, flags, "refused:", str(e)[:80]); continue
    want, ost, _ = orc.align_encoded(idx, seqs, lens, orc.params(flags), fresh_dseed=0)  # the serial reference
    ok = bw.aln_bytes(off, alns) == want
    st = ctx.stats()
    okc = (st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet) and st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes
    print("config", c, "len", ln, " ".join(flags), "->", "IDENTICAL" if ok else "MISMATCH", "" if okc else "(work counters differ)", flush=True)
    bad += (not ok) or (not okc)
print("failures:", bad)
sys.exit(1 if bad else 0)
