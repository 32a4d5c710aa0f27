"""Pins the CPU oracle (oracle/bwb_oracle.c) to golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from golden.make_golden import ALIGN_CONFIGS


def test_rank_known_answers(oracle, golden):
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"))
    pos = np.load(os.path.join(golden, "rank_pos.npy"))
    for inc in (0, 1):
        ref = np.load(os.path.join(golden, f"rank_O_alphabet_inc{inc}.npy"))
        got = oracle.O_alphabet(idx, pos, inc)
        assert np.array_equal(got[:, 1:], ref[:, 1:])
    ref = np.load(os.path.join(golden, "rank_O_single.npy"))
    sub = slice(0, 1500)
    assert np.array_equal(oracle.O_single(idx, pos[sub])[:, 1:], ref[sub, 1:])


def _read_fastq_codes(path):
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        lut[ord(ch)] = v
        lut[ord(ch.lower())] = v
    lines = open(path).read().split("\n")
    return [lut[np.frombuffer(lines[i].encode(), dtype=np.uint8)] for i in range(1, len(lines) - 1, 4)]


@pytest.mark.parametrize("tag", ["toy", "ragged"])
def test_calculate_d_known_answers(oracle, golden, tag):
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"))
    reads = _read_fastq_codes(os.path.join(golden, f"{tag}.fq"))
    vec = np.load(os.path.join(golden, f"dvec_{tag}.npy"))
    p = oracle.params()
    off = 0
    for r in reads[:150]:
        ln, seed = int(vec[off]), int(vec[off + 1]); off += 2
        assert ln == len(r)
        D = vec[off:off + 2 * (ln + 1)].reshape(ln + 1, 2); off += 2 * (ln + 1)
        Ds = vec[off:off + 2 * (seed + 1)].reshape(seed + 1, 2); off += 2 * (seed + 1)
        assert np.array_equal(oracle.calculate_d(idx, r, p), D)
        if ln > seed:
            assert np.array_equal(oracle.calculate_d(idx, r[:seed], p), Ds)


@pytest.mark.parametrize("name", sorted(ALIGN_CONFIGS))
def test_aln_bytes_match_reference(oracle, golden, tmp_path, name):
    out = str(tmp_path / "o.aln")
    p = oracle.params(ALIGN_CONFIGS[name])
    n, st, _ = oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, "toy.fq"), out, p)
    assert n == 600
    assert open(out, "rb").read() == open(os.path.join(golden, f"toy_{name}.aln"), "rb").read()


@pytest.mark.parametrize("name", ["n0", "n3", "n4gap", "s2", "p2", "n5"])
@pytest.mark.parametrize("threads", [1, 3])
def test_ragged_reads_match_reference(oracle, golden, tmp_path, name, threads):
    out = str(tmp_path / "o.aln")
    p = oracle.params(ALIGN_CONFIGS[name] + ["-t", str(threads)])
    n, _, _ = oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, "ragged.fq"), out, p)
    assert n == 200
    assert open(out, "rb").read() == open(os.path.join(golden, f"ragged_{name}.aln"), "rb").read()


@pytest.mark.parametrize("name,flags", [("n2", ["-n", "2"]), ("n3k1", ["-n", "3", "-k", "1"]), ("p2", ["-P", "-n", "2"])])
def test_short_reads_see_the_previous_long_reads_dseed(oracle, golden, tmp_path, name, flags):
    """short.fq mixes reads <= seed_length with longer ones; the reference was run with -t 1 (serial: one D_seed buffer for the
    whole file, inexact_match.c:35).  The oracle's serial mode must give those bytes; zeroing D_seed per read must not."""
    out = str(tmp_path / "o.aln")
    want = open(os.path.join(golden, f"short_{name}_t1.aln"), "rb").read()
    p = oracle.params(flags + ["-t", "1"])
    oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, "short.fq"), out, p, fresh_dseed=0)
    assert open(out, "rb").read() == want
    oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, "short.fq"), out, p, fresh_dseed=1)
    assert open(out, "rb").read() != want


HIGH_SCORE_FLAGS = ["-n", "5", "-M", "52", "-O", "60", "-E", "30"]


def test_scores_above_255_match_reference(oracle, golden, tmp_path):
    """himm.fq (4 % substitutions) with -n 5 -M 52: 642 heap buckets, five mismatches score 260.  The entry's 8-bit score field
    wraps (align.h:104; the break test inexact_match.c:309 sees it), the hit's score is the full int (:332,348): the reference's
    bytes hold scores above 255."""
    out = str(tmp_path / "o.aln")
    want = open(os.path.join(golden, "himm_n5bigpen.aln"), "rb").read()
    n, _, _ = oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, "himm.fq"), out, oracle.params(HIGH_SCORE_FLAGS))
    assert n == 160
    assert open(out, "rb").read() == want
    # (the fixture does exercise the range: some hit's score field exceeds 255)
    import struct
    data, pos, top = want, 0, 0
    while pos < len(data):
        (ne,) = struct.unpack_from("<i", data, pos); pos += 4
        for _ in range(ne):
            score, = struct.unpack_from("<i", data, pos)
            top = max(top, score)
            pairs, = struct.unpack_from("<i", data, pos + 4 + 16 + 16)
            pos += 4 + 16 + 16 + 4 + 4 * pairs
    assert top > 255


ROUND5_FIXTURES = [("sim_chr21_N100.fastq", "sim_chr21_N100_n0.aln", ["-n", "0"]), ("sim_chr21_N100.fastq", "sim_chr21_N100_n2.aln", ["-n", "2"]),
                   ("gapo.fq", "gapo_o6.aln", ["-n", "6", "-o", "6", "-e", "6", "-m", "200000"]), ("gapo.fq", "gapo_o5.aln", ["-n", "5", "-o", "5", "-e", "2"]),
                   ("himm.fq", "himm_M80.aln", ["-n", "3", "-M", "80", "-O", "90", "-E", "70"])]


@pytest.mark.parametrize("fq,aln,flags", ROUND5_FIXTURES)
def test_round5_fixtures_match_reference(oracle, golden, tmp_path, fq, aln, flags):
    """the reference's own test input (test_data/sim_chr21_N100.fastq: config C1) on the toy index; alignments with up to six gap opens
    (gapo.fq: reads with 2..6 single-base indels); penalties above 63 (990 heap buckets)"""
    out = str(tmp_path / "o.aln")
    oracle.align_fastq(os.path.join(golden, "toy.fa.bwt"), os.path.join(golden, fq), out, oracle.params(flags))
    assert open(out, "rb").read() == open(os.path.join(golden, aln), "rb").read()
