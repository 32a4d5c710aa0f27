"""GPU parity tests: the HIP path (through the C-ABI, bwbble_amd ctypes mirror) against the golden vectors of
the real reference and against the CPU oracle on the same seeded inputs.  Bit-exact (integer work)."""
import os

import numpy as np
import pytest

import bwbble_amd as bw
from golden.make_golden import ALIGN_CONFIGS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy_ctx(golden, built):
    ctx = bw.Context(os.path.join(golden, "toy.fa.bwt"))
    yield ctx
    ctx.close()


def test_rank16_matches_reference_O_alphabet(toy_ctx, golden):
    pos = np.load(os.path.join(golden, "rank_pos.npy"))
    for inc in (0, 1):
        ref = np.load(os.path.join(golden, f"rank_O_alphabet_inc{inc}.npy"))
        got = toy_ctx.rank16(pos, inc=inc, exact=False)
        assert np.array_equal(got[:, 1:], ref[:, 1:])


def test_rank16_exact_matches_reference_O(toy_ctx, golden):
    pos = np.load(os.path.join(golden, "rank_pos.npy"))
    ref = np.load(os.path.join(golden, "rank_O_single.npy"))
    got = toy_ctx.rank16(pos, inc=0, exact=True)
    Cj = toy_ctx.bwt.C[:16].astype(np.uint64)
    assert np.array_equal(got[:, 1:], ref[:, 1:] + Cj[None, 1:])


@pytest.mark.parametrize("tag", ["toy", "ragged"])
def test_calculate_d_matches_reference(toy_ctx, golden, tag):
    reads = bw.read_fastq(os.path.join(golden, f"{tag}.fq"))
    seqs, lens = bw.encode_reads(reads)
    p = bw.params()
    D, Ds = toy_ctx.calc_d(p, seqs, lens)
    vec = np.load(os.path.join(golden, f"dvec_{tag}.npy"))
    off = 0
    for i in range(len(reads)):
        ln, seed = int(vec[off]), int(vec[off + 1]); off += 2
        rD = vec[off:off + 2 * (ln + 1)].reshape(ln + 1, 2); off += 2 * (ln + 1)
        rS = vec[off:off + 2 * (seed + 1)].reshape(seed + 1, 2); off += 2 * (seed + 1)
        assert np.array_equal(D[i, :ln + 1], rD), f"D mismatch read {i}"
        if ln > seed:
            assert np.array_equal(Ds[i], rS), f"D_seed mismatch read {i}"


@pytest.mark.parametrize("name", sorted(ALIGN_CONFIGS))
def test_aln_bytes_match_reference_toy(toy_ctx, golden, name):
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "toy.fq")))
    off, alns = toy_ctx.align(bw.params(ALIGN_CONFIGS[name]), seqs, lens)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, f"toy_{name}.aln"), "rb").read()


@pytest.mark.parametrize("name", ["n0", "n3", "n4gap", "s2", "p2"])
def test_aln_bytes_match_reference_ragged(toy_ctx, golden, name):
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "ragged.fq")))
    off, alns = toy_ctx.align(bw.params(ALIGN_CONFIGS[name]), seqs, lens)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, f"ragged_{name}.aln"), "rb").read()


@pytest.mark.parametrize("name,flags", [("n2", ["-n", "2"]), ("n3k1", ["-n", "3", "-k", "1"]), ("p2", ["-P", "-n", "2"])])
def test_short_reads_match_serial_reference(toy_ctx, golden, name, flags):
    """reads <= seed_length between longer ones: the bytes of the reference's serial path (-t 1), SURVEY Appendix B-11"""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "short.fq")))
    off, alns = toy_ctx.align(bw.params(flags), seqs, lens)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, f"short_{name}_t1.aln"), "rb").read()


@pytest.mark.parametrize("name", ["n0", "n3"])
def test_work_counters_match_oracle(toy_ctx, oracle, golden, name):
    """The visit counts that feed roofline.achieved are the SURVEY 8(d) algorithmic counts."""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "toy.fq")))
    toy_ctx.align(bw.params(ALIGN_CONFIGS[name]), seqs, lens)
    st = toy_ctx.stats()
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"))
    _, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(ALIGN_CONFIGS[name]))
    assert st.visits_single == ost.visits_single
    assert st.visits_alphabet == ost.visits_alphabet
    assert st.heap_pops == ost.heap_pops
    assert st.heap_pushes == ost.heap_pushes
    assert st.n_alignments == ost.n_alignments


def test_empty_and_degenerate_batches(toy_ctx):
    p = bw.params(["-n", "2"])
    off, alns = toy_ctx.align(p, np.zeros((0, 1), dtype=np.uint8), np.zeros(0, dtype=np.uint16))
    assert len(off) == 1 and len(alns) == 0
    # all-N read, and a read with more N's than max_diff: empty records (inexact_match.c:260-266)
    seqs = np.full((2, 40), 4, dtype=np.uint8)
    seqs[1, :37] = 0
    off, alns = toy_ctx.align(p, seqs, np.array([40, 40], dtype=np.uint16))
    assert list(off) == [0, 0, 0]


def test_unsupported_parameters_fail_loudly(toy_ctx):
    seqs = np.zeros((1, 40), dtype=np.uint8)
    lens = np.array([40], dtype=np.uint16)
    with pytest.raises(bw.BwbError):
        toy_ctx.align(bw.params(["-o", "9"]), seqs, lens)
    with pytest.raises(bw.BwbError):
        toy_ctx.align(bw.params(["-n", "200"]), seqs, lens)
    with pytest.raises(bw.BwbError):  # result of a slot that was never submitted
        toy_ctx.slot_result(3)


def test_scores_above_255_match_reference(toy_ctx, golden):
    """-n 5 -M 52 -O 60 -E 30 on mismatch-rich reads: hits that score above 255 (bwb_aln.score is 16 bits wide since ABI version 2)
    and the 8-bit wrap of the entry's score in the break test (inexact_match.c:309) - byte-identical to the REAL reference."""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "himm.fq")))
    off, alns = toy_ctx.align(bw.params(["-n", "5", "-M", "52", "-O", "60", "-E", "30"]), seqs, lens)
    assert int(alns["score"].max()) > 255
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, "himm_n5bigpen.aln"), "rb").read()


@pytest.mark.parametrize("aln,flags", [("gapo_o5.aln", ["-n", "5", "-o", "5", "-e", "2"]), ("gapo_o6.aln", ["-n", "6", "-o", "6", "-e", "6", "-m", "200000"])])
def test_more_than_four_gap_opens_match_reference(toy_ctx, oracle, golden, aln, flags):
    """-o 5 and -o 6 (round 5, ABI version 3: a heap entry and a bwb_aln hold eight gap runs): reads with 2..6 single-base insertions or
    deletions, whose best alignments open that many gaps - byte-identical to the REAL reference (tests/golden/make_golden.py round5),
    work counters equal to the oracle's; -o 9 is refused."""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "gapo.fq")))
    off, alns = toy_ctx.align(bw.params(flags), seqs, lens)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, aln), "rb").read()
    assert int(alns["num_gapo"].max()) >= 5
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"))
    _, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags))
    st = toy_ctx.stats()
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes and st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet
    with pytest.raises(bw.BwbError):
        toy_ctx.align(bw.params(["-n", "9", "-o", "9"]), seqs[:1], lens[:1])


def test_penalty_and_score_range_limits(toy_ctx, oracle, golden):
    """Score ranges beyond round 2's 128 heap buckets work (bucket-state rows are sized by the range: here (n+1) M + 2 O + 7 E = 320
    buckets) and equal the oracle.  Penalties above 63 (round 5): a child's bucket can lie beyond the 64-bucket window of the non-empty
    buckets - the 32-byte-entry kernels handle that (LHeap::far) - with the REAL reference's bytes for -M 80 -O 90 -E 70 (990 buckets)
    and the oracle's for mixed sizes; above 255, or more than 1 024 buckets, is refused loudly."""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "toy.fq"), max_reads=300))
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"))
    for flags in (["-n", "3", "-M", "40", "-O", "45", "-E", "10"], ["-n", "3", "-M", "70", "-O", "65", "-E", "5"], ["-n", "2", "-M", "3", "-O", "200", "-E", "100", "-e", "3"],
                  ["-n", "4", "-M", "130", "-O", "11", "-E", "64", "-e", "2"], ["-n", "1", "-M", "255", "-O", "255", "-E", "2", "-o", "0"]):
        off, alns = toy_ctx.align(bw.params(flags), seqs, lens)
        want, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags))
        assert bw.aln_bytes(off, alns) == want, flags
        st = toy_ctx.stats()
        assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes, flags
    hs, hl = bw.encode_reads(bw.read_fastq(os.path.join(golden, "himm.fq")))
    off, alns = toy_ctx.align(bw.params(["-n", "3", "-M", "80", "-O", "90", "-E", "70"]), hs, hl)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, "himm_M80.aln"), "rb").read()
    for bad in (["-M", "256"], ["-O", "300"], ["-E", "1000"], ["-n", "100", "-M", "60"], ["-n", "3", "-M", "200", "-O", "200", "-E", "200"]):
        with pytest.raises(bw.BwbError):
            toy_ctx.align(bw.params(bad), seqs[:1], lens[:1])
    toy_ctx.align(bw.params(["-n", "2"]), seqs[:50], lens[:50])  # the context is still usable
