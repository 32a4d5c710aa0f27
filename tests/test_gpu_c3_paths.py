"""The code paths that only a GRCh37-scale index (config C3: 6.85 G BWT rows) reaches, on a 6 M-row index:

* superblock rows >= 1 of the base table (bwb_device.h): the TEST build of the library (`make testlib`) uses 2^13-block
  superblocks, so the mid genome spans 7 of them like C3 spans 4 of 2^24 blocks;
* 64-bit positions with their three high bits packed into the last word of a 16-byte heap entry (bwb_lane.h LHeap::pack_w):
  the test build stores positions + 5 * 2^32, and BWB_FORCE_POS64 selects the 64-bit kernels;
* config C5 of BASELINE.json exactly: 150 bp reads, -n 5 -o 1 -e 6 -l 32 -k 2 (the last four are the defaults, align.c:26-30).

Bit-exact against the oracle (pinned to the reference by tests/test_oracle_golden.py) and the reference's golden files."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bwbble_amd as bw
from golden.make_golden import ALIGN_CONFIGS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mid(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("c3mid")
    fa = str(d / "g.fa")
    subprocess.run([bw.SYNTH_BIN, "genome", fa, "3000000", "5", "1200", "77"], check=True)
    subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL)
    return d, fa


@pytest.mark.parametrize("extra_env", [{}, {"BWB_SLICE_ITERS": "120"}, {"BWB_DTAB": "1", "BWB_DTAB_K": "9"}])  # (the last: the calculate_d table across superblock rows)
def test_superblock_rows_and_packed_high_bits(mid, extra_env):
    d, fa = mid
    bw.build(testlib=True)
    env = dict(os.environ, BWB_LIB=bw.TEST_LIB_PATH, BWB_FORCE_POS64="1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "c3_paths_worker.py"), fa, str(d)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "C3-PATHS-OK 5 configurations" in r.stdout, r.stdout[-3000:]


def test_config_c5_matches_oracle(mid, oracle):
    """150 bp reads, 1 % substitutions + indel reads, -n 5 -o 1 -e 6 -l 32 -k 2"""
    d, fa = mid
    fq = str(d / "c5.fq")
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, "700", "150", "55", "1.0", "20.0", "0.5"], check=True)
    seqs, lens = bw.load_fastq_codes(fq)
    flags = ["-n", "5", "-o", "1", "-e", "6", "-l", "32", "-k", "2"]
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    off, alns = ctx.align(bw.params(flags), seqs, lens)
    want, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags))
    assert bw.aln_bytes(off, alns) == want
    st = ctx.stats()
    assert st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes
    assert sum(1 for a in alns if a["num_gapo"]) > 0  # gapped hits are present
    ctx.close()


def test_config_c5_matches_reference_golden(golden):
    """the reference itself on the ragged reads (36..150 bp) with -n 5 and the default -o/-e/-l/-k: tests/golden/ragged_n5.aln"""
    seqs, lens = bw.encode_reads(bw.read_fastq(os.path.join(golden, "ragged.fq")))
    ctx = bw.Context(os.path.join(golden, "toy.fa.bwt"))
    off, alns = ctx.align(bw.params(ALIGN_CONFIGS["n5"]), seqs, lens)
    assert bw.aln_bytes(off, alns) == open(os.path.join(golden, "ragged_n5.aln"), "rb").read()
    ctx.close()
