"""One BASELINE config at its REAL index size inside the GPU test session: config C2 - a chr21-scale synthetic
multi-genome of 48 M forward characters (106 M BWT rows) built by the product's own indexer, 20 000 x 100 bp reads, `-n 3`
and the CLI default `-n 0` - through the product library (not the small-superblock test build), compared byte for byte and
counter for counter with the CPU oracle, and with the real reference binary (oracle/_ref/bwbble, built in the container and
shipped with the snapshot) when it is there.  About a minute on the GPU box."""
import os
import subprocess

import pytest

import bwbble_amd as bw

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bwbble")
N_FWD, N_READS = 48_000_000, 20_000


@pytest.fixture(scope="module")
def c2(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("c2")
    fa, fq = str(d / "c2.fa"), str(d / "c2.fq")
    # the same generator calls as bench.py --genome-mb 48 (one record, 20 000 bubbles, seed 21)
    subprocess.run([bw.SYNTH_BIN, "genome", fa, str(N_FWD), "1", str(N_FWD // 2400), "21"], check=True)
    subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(N_READS), "100", "1000", "1.0", "0.1", "0.0"], check=True)
    ctx = bw.Context(fa + ".bwt")
    yield d, fa, fq, ctx
    ctx.close()


@pytest.mark.parametrize("flags", [["-n", "3"], ["-n", "0"]])
def test_c2_index_matches_oracle_and_reference(c2, oracle, flags):
    d, fa, fq, ctx = c2
    assert ctx.bwt.length > 100_000_000  # the real C2 size: 2 x (48 M + bubbles) + separators
    seqs, lens = bw.load_fastq_codes(fq)
    off, alns = ctx.align(bw.params(flags), seqs, lens)
    got = bw.aln_bytes(off, alns)
    st = ctx.stats()
    idx = oracle.load_index(fa + ".bwt")
    want, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags + ["-t", str(os.cpu_count() or 1)]))
    assert got == want
    assert st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes and st.n_alignments == ost.n_alignments
    assert int(off[-1]) > (N_READS // 2 if flags[1] != "0" else N_READS // 4)  # most reads map with -n 3; 0.99^100 = 37 % are error-free
    if os.path.exists(REF_BIN):
        out = str(d / ("ref" + "".join(flags) + ".aln"))
        subprocess.run([REF_BIN, "align"] + flags + ["-t", str(os.cpu_count() or 1), fa, fq, out], check=True, stdout=subprocess.DEVNULL)
        assert got == open(out, "rb").read()
