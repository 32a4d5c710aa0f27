"""Parity at GRCh37 INDEX SIZE inside the GPU test suite: 64-bit positions, several superblock rows, bucket
loads from HBM, slots reused and reads parked - on the product library, against the real reference binary when it travelled with the
snapshot (oracle/_ref/bwbble), else against the pinned restatement (oracle/libbwb_oracle.so).

The genome (3.1 G forward characters -> 6.85 G BWT rows) and its index are built by the product's own tools into bench.py's work
directory under bench.py's file names, so a bench run on the same box finds them.  These are the only parity tests at C3 / C5 size, so
they do not skip by themselves: a box that cannot build the index (less than 120 GB of available memory, or slower than the
build budget) FAILS them with the reason, and only an explicit BWB_SKIP_GRCH37=1 skips.  The build time is printed.  (Runs last: the
file name sorts after the other GPU tests.)"""
import os
import subprocess
import time

import numpy as np
import pytest

import bwbble_amd as bw

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FWD = int(os.environ.get("BWB_TEST_GRCH37_FWD", 3_100_000_000))
WORK = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
BUILD_BUDGET_S = int(os.environ.get("BWB_TEST_GRCH37_BUILD_BUDGET_S", 600))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bwbble")


def mem_available_gb():
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable:"):
            return int(ln.split()[1]) / 1e6
    return 0.0


@pytest.fixture(scope="module")
def grch37(built):
    if os.environ.get("BWB_SKIP_GRCH37"):
        pytest.skip("BWB_SKIP_GRCH37 is set")
    fa = os.path.join(WORK, f"genome_{N_FWD}.fa")
    ok = fa + ".bwt.ok"
    if not os.path.exists(ok):
        if mem_available_gb() < 120 * (N_FWD / 3.1e9):
            pytest.fail(f"building a {N_FWD / 1e9:.1f} G-character index needs ~120 GB of host memory, {mem_available_gb():.0f} GB available "
                        "(set BWB_SKIP_GRCH37=1 to skip the GRCh37-size parity tests on purpose)")
        os.makedirs(WORK, exist_ok=True)
        t0 = time.time()
        try:
            subprocess.run([bw.SYNTH_BIN, "genome", fa, str(N_FWD), "24" if N_FWD > 60_000_000 else "1", str(max(4, N_FWD // 2400)), "21"], check=True, timeout=BUILD_BUDGET_S)
            subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL, timeout=max(30, BUILD_BUDGET_S - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            pytest.fail(f"building the GRCh37-scale index took longer than {BUILD_BUDGET_S} s on this box (BWB_TEST_GRCH37_BUILD_BUDGET_S raises the "
                        "budget, BWB_SKIP_GRCH37=1 skips on purpose)")
        if os.path.exists(fa + ".ref"):
            os.remove(fa + ".ref")
        open(ok, "w").write("ok\n")
        print(f"[grch37] genome + index of {N_FWD} forward characters built in {time.time() - t0:.0f} s")
    else:
        print("[grch37] index found in " + WORK)
    return fa


def reference_bytes(fa, fq, flags, tmp, oracle):
    """.aln bytes of the reads in fq: the real reference when its binary is here, else the pinned restatement; plus the
    restatement's work counters (None with the real reference, which does not count)"""
    out = os.path.join(tmp, "ref.aln")
    threads = str(min(64, os.cpu_count() or 8))
    if os.path.exists(REF_BIN) and not os.environ.get("BWB_TEST_USE_ORACLE"):
        subprocess.run([REF_BIN, "align"] + flags + ["-t", threads, fa, fq, out], check=True, stdout=subprocess.DEVNULL)
        return open(out, "rb").read(), None, "oracle/_ref/bwbble"
    n, st, _ = oracle.align_fastq(fa + ".bwt", fq, out, oracle.params(flags + ["-t", threads]))
    return open(out, "rb").read(), st, "libbwb_oracle.so"


def test_c3_tail_of_a_long_stream_matches_the_reference(grch37, oracle, tmp_path):
    """Config C3's parameters on its real index: 2.6 M reads streamed through the eight slots in chunks of 200 k (every slot is
    reused, every slice parks reads); the LAST 20 000 reads' records - produced while slots 0..4 are in their second use and the
    earlier chunks' heavy reads are still being resumed - must be the reference's bytes; and the same 20 000 reads aligned alone
    must give the same bytes and the restatement's work counters."""
    fa = grch37
    n_stream, chunk, n_tail = 2_600_000, 200_000, 20_000
    fq = str(tmp_path / "stream.fq")
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(n_stream), "100", "4242", "1.0", "0.1", "0.0"], check=True)
    seqs, lens = bw.load_fastq_codes(fq)
    tail_fq = str(tmp_path / "tail.fq")
    with open(fq) as f, open(tail_fq, "w") as g:  # the last n_tail records of the stream as their own FASTQ
        lines = f.readlines()
        g.writelines(lines[-4 * n_tail:])
    del lines
    flags = ["-n", "3"]
    p = bw.params(flags)
    ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
    assert not ctx.bwt.length < 0xFFFFFFFF or N_FWD < 2_000_000_000  # 64-bit positions at the real size
    nchunks = n_stream // chunk
    results = {}
    t0 = time.time()
    for c in range(nchunks):
        slot = c % bw.MAX_SLOTS
        if c >= bw.MAX_SLOTS:
            results[c - bw.MAX_SLOTS] = ctx.slot_result(slot)
        ctx.slot_upload(slot, p, seqs[c * chunk:(c + 1) * chunk], lens[c * chunk:(c + 1) * chunk])
        ctx.slot_submit(slot)
    for c in range(max(0, nchunks - bw.MAX_SLOTS), nchunks):
        results[c] = ctx.slot_result(c % bw.MAX_SLOTS)
    ctx.flush()
    st = ctx.stats()
    assert st.n_parked_reads > 0 and st.launches_search >= nchunks
    off, alns = results[nchunks - 1]
    first = chunk - n_tail
    streamed = bw.aln_bytes(off[first:] - off[first], alns[int(off[first]):int(off[-1])])
    print(f"[grch37] streamed {n_stream} reads in {time.time() - t0:.1f} s, {st.n_parked_reads} reads parked")
    want, ost, who = reference_bytes(fa, tail_fq, flags, str(tmp_path), oracle)
    assert streamed == want, f"the stream's last {n_tail} records differ from {who}"
    # the same reads alone: one batch, work counters
    ctx.reset_stats()
    off1, alns1 = ctx.align(p, seqs[-n_tail:], lens[-n_tail:])
    assert bw.aln_bytes(off1, alns1) == want
    if ost is None:  # (the real reference does not count: the restatement does, on the last 2 000 reads)
        _, ost, _ = oracle.align_encoded(oracle.load_index(fa + ".bwt"), seqs[-2000:], lens[-2000:], oracle.params(flags + ["-t", str(min(64, os.cpu_count() or 8))]))
        ctx.reset_stats()
        ctx.align(p, seqs[-2000:], lens[-2000:])
    st1 = ctx.stats()
    assert st1.heap_pops == ost.heap_pops and st1.heap_pushes == ost.heap_pushes
    assert st1.visits_single + st1.visits_alphabet == ost.visits_single + ost.visits_alphabet
    ctx.close()


def test_c5_reads_match_the_reference_at_grch37_size(grch37, oracle, tmp_path):
    """Config C5's parameters (150 bp reads with indels, -n 5 -o 1 -e 6 -l 32 -k 2) on the GRCh37-scale index: 2 000 reads, bytes
    equal to the reference's."""
    fa = grch37
    fq = str(tmp_path / "c5.fq")
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, "2000", "150", "5151", "1.0", "0.2", "0.0"], check=True)
    seqs, lens = bw.load_fastq_codes(fq)
    flags = ["-n", "5", "-o", "1", "-e", "6", "-l", "32", "-k", "2"]
    ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
    off, alns = ctx.align(bw.params(flags), seqs, lens)
    want, _, who = reference_bytes(fa, fq, flags, str(tmp_path), oracle)
    assert bw.aln_bytes(off, alns) == want, f"differs from {who}"
    ctx.close()


C5_FLAGS = ["-n", "5", "-o", "1", "-e", "6", "-l", "32", "-k", "2"]


def stream_chunks(ctx, p, seqs, lens, chunk):
    """every chunk through the context's slots in order (slots are reused, slices park reads); returns the per-chunk results"""
    nchunks = len(lens) // chunk
    results = {}
    for c in range(nchunks):
        slot = c % bw.MAX_SLOTS
        if c >= bw.MAX_SLOTS:
            results[c - bw.MAX_SLOTS] = ctx.slot_result(slot)
        ctx.slot_upload(slot, p, seqs[c * chunk:(c + 1) * chunk], lens[c * chunk:(c + 1) * chunk])
        ctx.slot_submit(slot)
    for c in range(max(0, nchunks - bw.MAX_SLOTS), nchunks):
        results[c] = ctx.slot_result(c % bw.MAX_SLOTS)
    ctx.flush()
    return results


def test_c5_stream_and_rerun_paths_match_the_reference_at_grch37_size(grch37, oracle, tmp_path, monkeypatch):
    """Config C5 as a STREAM at GRCh37 size: 400 000 reads of 150 bp through the eight slots in chunks of 40 000
    (slots reused, reads parked and resumed with 64-bit positions in their entries); the last 5 000 records must be the reference's
    bytes.  Then the pool-exhaustion path at this size: a context with a 1 GB chunk pool gives reads up and re-runs them in the larger
    scratch classes - same bytes - with 16-byte entries (-o 1) and with 32-byte entries (-o 2)."""
    fa = grch37
    n_stream, chunk, n_tail = 400_000, 40_000, 5_000
    fq = str(tmp_path / "c5s.fq")
    subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(n_stream), "150", "6161", "1.0", "0.2", "0.0"], check=True)
    seqs, lens = bw.load_fastq_codes(fq)
    tail_fq = str(tmp_path / "c5tail.fq")
    with open(fq) as f, open(tail_fq, "w") as g:
        lines = f.readlines()
        g.writelines(lines[-4 * n_tail:])
    del lines
    p = bw.params(C5_FLAGS)
    ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
    t0 = time.time()
    results = stream_chunks(ctx, p, seqs, lens, chunk)
    st = ctx.stats()
    assert st.n_parked_reads > 0 and st.launches_search >= n_stream // chunk
    off, alns = results[n_stream // chunk - 1]
    first = chunk - n_tail
    streamed = bw.aln_bytes(off[first:] - off[first], alns[int(off[first]):int(off[-1])])
    print(f"[grch37] C5: streamed {n_stream} reads in {time.time() - t0:.1f} s, {st.n_parked_reads} reads parked, {st.n_overflow_reads} re-run")
    want, _, who = reference_bytes(fa, tail_fq, C5_FLAGS, str(tmp_path), oracle)
    assert streamed == want, f"the C5 stream's last {n_tail} records differ from {who}"
    ctx.close()
    # the re-run path at this size: a pool that cannot hold the heaps of the reads in flight
    monkeypatch.setenv("BWB_POOL_GB", "1")
    ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
    off1, alns1 = ctx.align(p, seqs[-n_tail:], lens[-n_tail:])
    st1 = ctx.stats()
    print(f"[grch37] C5 with a 1 GB pool: {st1.n_overflow_reads} of {n_tail} reads re-run")
    assert st1.n_overflow_reads > 0
    assert bw.aln_bytes(off1, alns1) == want, f"re-run reads (16-byte entries) differ from {who}"
    # 32-byte entries (-o 2), streamed in four chunks with the small pool, against the reference on the same 2 000 reads
    n2 = 2_000
    flags2 = ["-n", "5", "-o", "2", "-e", "6", "-l", "32", "-k", "2"]
    fq2 = str(tmp_path / "c5o2.fq")
    with open(tail_fq) as f, open(fq2, "w") as g:
        g.writelines(f.readlines()[-4 * n2:])
    ctx.reset_stats()
    res2 = stream_chunks(ctx, bw.params(flags2), seqs[-n2:], lens[-n2:], n2 // 4)
    st2 = ctx.stats()
    got2 = b"".join(bw.aln_bytes(res2[c][0], res2[c][1]) for c in range(4))
    want2, _, who2 = reference_bytes(fa, fq2, flags2, str(tmp_path), oracle)
    print(f"[grch37] C5 -o 2 with a 1 GB pool: {st2.n_overflow_reads} of {n2} reads re-run")
    assert got2 == want2, f"-o 2 (32-byte entries) differs from {who2}"
    ctx.close()
