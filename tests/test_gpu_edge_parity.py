"""GPU parity on edge cases and on the overflow / re-run machinery, against the CPU oracle on the same seeded inputs
(the oracle itself is pinned to the real reference by tests/test_oracle_golden.py).  Bit-exact .aln bytes."""
import os
import subprocess

import numpy as np
import pytest

import bwbble_amd as bw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mid(built, tmp_path_factory):
    """3 M-character synthetic multi-genome (repeats, SNP codes, bubbles) indexed by the product's own indexer."""
    d = tmp_path_factory.mktemp("mid")
    fa = str(d / "g.fa")
    subprocess.run([bw.SYNTH_BIN, "genome", fa, "3000000", "5", "1200", "77"], check=True)
    subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL)
    return d, fa


def synth_reads(fa, path, n, length, seed, sub=1.0, indel=0.1, npct=0.0):
    subprocess.run([bw.SYNTH_BIN, "reads", fa, path, str(n), str(length), str(seed), str(sub), str(indel), str(npct)], check=True)
    return bw.load_fastq_codes(path)


def check(ctx, oracle, idx, flags, seqs, lens, fresh=1):
    off, alns = ctx.align(bw.params(flags), seqs, lens)
    want, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags), fresh_dseed=fresh)
    assert bw.aln_bytes(off, alns) == want
    st = ctx.stats()
    assert st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes
    return st


@pytest.fixture(scope="module")
def mid_ctx(mid, oracle):
    d, fa = mid
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    yield d, fa, ctx, idx
    ctx.close()


@pytest.mark.parametrize("flags", [["-n", "0"], ["-n", "2"], ["-n", "3"], ["-n", "4", "-o", "2", "-e", "4"],
                                   ["-n", "3", "-l", "0"], ["-n", "3", "-k", "0", "-l", "40"],
                                   ["-S", "-n", "0"], ["-S", "-n", "3"], ["-S", "-n", "4", "-o", "2", "-e", "4"],
                                   ["-P", "-n", "0"], ["-P", "-n", "3"], ["-P", "-S", "-n", "2"]])
def test_mid_genome_matches_oracle(mid_ctx, oracle, flags):
    d, fa, ctx, idx = mid_ctx
    seqs, lens = synth_reads(fa, str(d / "a.fq"), 3000, 100, 5, sub=1.5, indel=2.0, npct=2.0)
    check(ctx, oracle, idx, flags, seqs, lens)


def test_max_entries_break(mid_ctx, oracle):
    """-m small: the search stops when the heap holds more than max_entries entries (inexact_match.c:299)."""
    d, fa, ctx, idx = mid_ctx
    seqs, lens = synth_reads(fa, str(d / "m.fq"), 1500, 100, 6, sub=2.0)
    for m in ("40", "500"):
        check(ctx, oracle, idx, ["-n", "3", "-m", m], seqs, lens)


def test_short_and_long_reads(mid_ctx, oracle):
    """Reads shorter than the seed (D_seed defined as zeros), at the seed length, and 250-base reads."""
    d, fa, ctx, idx = mid_ctx
    for ln, n in ((20, 400), (32, 400), (33, 400), (250, 300)):
        seqs, lens = synth_reads(fa, str(d / f"l{ln}.fq"), n, ln, 10 + ln, sub=1.0, indel=1.0, npct=3.0)
        check(ctx, oracle, idx, ["-n", "2"], seqs, lens)


def test_ragged_batch_with_n_rich_reads(mid_ctx, oracle):
    d, fa, ctx, idx = mid_ctx
    parts = [synth_reads(fa, str(d / f"r{ln}.fq"), 300, ln, 40 + ln, sub=2.0, indel=3.0, npct=30.0) for ln in (36, 70, 101, 150)]
    stride = max(p[0].shape[1] for p in parts)
    seqs = np.full((sum(len(p[1]) for p in parts), stride), 4, dtype=np.uint8)
    lens = np.concatenate([p[1] for p in parts])
    o = 0
    for s_, l_ in parts:
        seqs[o:o + len(l_), :s_.shape[1]] = s_
        o += len(l_)
    seqs[5, :lens[5]] = 4   # an all-N read
    seqs[6, 3:9] = 4        # more N's than max_diff
    check(ctx, oracle, idx, ["-n", "3"], seqs, lens)


def test_scratch_overflow_classes_are_exact(mid, oracle, monkeypatch):
    """A tiny heap pool forces reads through the class-1/2 re-run path (still on the GPU): results must not change."""
    d, fa = mid
    monkeypatch.setenv("BWB_POOL_GB", "0")  # clamps to the 64 MB floor: 65 536 chunks, not enough for 5 000 reads at once
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "o.fq"), 5000, 100, 8, sub=2.5)
    st = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    assert st.n_overflow_reads > 0
    ctx.close()


def test_heavy_read_pass_is_exact(mid, oracle, monkeypatch):
    """BWB_ITER_BUDGET parks long-running reads and restarts them in the one-read-per-octet pass."""
    d, fa = mid
    monkeypatch.setenv("BWB_ITER_BUDGET", "1500")
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "h.fq"), 4000, 100, 9, sub=2.0)
    st = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    assert st.n_heavy_reads > 0
    ctx.close()


def test_force_64bit_positions(mid, oracle, monkeypatch):
    """The 64-bit position instantiations (used for > 4 G-row indexes; 16-byte entries, 32-byte ones for -o > 1) on a small index."""
    d, fa = mid
    monkeypatch.setenv("BWB_FORCE_POS64", "1")
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "w.fq"), 2000, 100, 12, sub=1.5, indel=2.0)
    check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    check(ctx, oracle, idx, ["-n", "3", "-o", "2"], seqs, lens)
    check(ctx, oracle, idx, ["-S", "-n", "2"], seqs, lens)
    check(ctx, oracle, idx, ["-P", "-n", "2"], seqs, lens)
    ctx.close()


def test_random_parameter_sweep(mid_ctx, oracle):
    """Seeded random combinations of flags, read lengths and error rates (the long version is tools/fuzz_parity.py)."""
    import random
    d, fa, ctx, idx = mid_ctx
    rng = random.Random(20240611)
    for c in range(10):
        ln = rng.choice([24, 36, 50, 76, 100, 125, 150, 200])
        seqs, lens = synth_reads(fa, str(d / f"z{c}.fq"), rng.choice([300, 800]), ln, rng.randrange(10000),
                                 sub=rng.choice([0.5, 1.0, 2.0, 4.0]), indel=rng.choice([0.0, 1.0, 5.0]), npct=rng.choice([0.0, 1.0, 10.0]))
        flags = ["-n", str(rng.choice([0, 1, 2, 3, 4])), "-o", str(rng.choice([0, 1, 2, 3])), "-e", str(rng.choice([0, 2, 6])),
                 "-l", str(rng.choice([0, 16, 32, 60])), "-k", str(rng.choice([0, 1, 2, 3])), "-M", str(rng.choice([1, 3, 5])),
                 "-O", str(rng.choice([3, 11])), "-E", str(rng.choice([1, 4])), "-m", str(rng.choice([200, 5000, 3000000]))]
        if rng.random() < 0.25:
            flags.append("-S")
        if rng.random() < 0.25:
            flags.append("-P")
        check(ctx, oracle, idx, flags, seqs, lens)
