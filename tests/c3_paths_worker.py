"""Worker of tests/test_gpu_c3_paths.py: runs in its own process so that bwbble_amd binds the TEST build of the library
(BWB_LIB=bwbble_amd/libbwbble_hip_test.so: 2^13-block superblocks, biased stored positions) instead of the product's.
usage: c3_paths_worker.py <genome.fa (indexed)> <workdir>"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bwbble_amd as bw  # noqa: E402
import oracle_lib  # noqa: E402

fa, work = sys.argv[1], sys.argv[2]
assert os.path.samefile(bw.LIB_PATH, bw.TEST_LIB_PATH), "the worker must run on the test build"
orc = oracle_lib.load()
idx = orc.load_index(fa + ".bwt")
ctx = bw.Context(fa + ".bwt")
nblk = (ctx.bwt.length + 127) // 128
assert nblk > 5 * 8192, "the index must span several 2^13-block superblocks"

# rank across superblock boundaries: O_alphabet and exact Occ16 against the oracle
rng = np.random.default_rng(11)
pos = [2**64 - 1, ctx.bwt.length - 1, 0]
for sb in range(1, nblk // 8192 + 1):
    b = sb * 8192 * 128
    pos += [p for p in (b - 129, b - 128, b - 1, b, b + 1, b + 127, b + 128) if p < ctx.bwt.length]
pos += [int(v) for v in rng.integers(0, ctx.bwt.length, 4000)]
pos = np.array(pos, dtype=np.uint64)
for inc in (0, 1):
    assert np.array_equal(ctx.rank16(pos, inc=inc)[:, 1:], orc.O_alphabet(idx, pos, inc)[:, 1:])
Cj = ctx.bwt.C[:16].astype(np.uint64)
sub = pos[:600]
assert np.array_equal(ctx.rank16(sub, inc=0, exact=True)[:, 1:], orc.O_single(idx, sub)[:, 1:] + Cj[None, 1:])

fq = os.path.join(work, "c3w.fq")
subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, "2500", "100", "12", "1.5", "2.0", "1.0"], check=True)
seqs, lens = bw.load_fastq_codes(fq)
n_checked = 0
for flags in (["-n", "3"], ["-n", "3", "-o", "2"], ["-S", "-n", "2"], ["-P", "-n", "2"], ["-n", "0"]):
    off, alns = ctx.align(bw.params(flags), seqs, lens)
    want, ost, _ = orc.align_encoded(idx, seqs, lens, orc.params(flags))
    assert bw.aln_bytes(off, alns) == want, flags
    st = ctx.stats()
    assert st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet, flags
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes, flags
    n_checked += 1
# SA lookups walk across superblocks too
b = bw.BwtFile(fa + ".bwt", load_sa=True)
ctx.set_sa(b.SA)
idx_sa = orc.load_index(fa + ".bwt", load_sa=True)
rows = rng.integers(0, b.length, 1500).astype(np.uint64)
assert np.array_equal(ctx.locate(rows), np.array([orc.lib.bwb_or_SA(idx_sa, int(r)) for r in rows], dtype=np.uint64))
ctx.close()
print(f"C3-PATHS-OK {n_checked} configurations, {len(pos)} rank positions, superblocks {nblk // 8192 + 1}")
