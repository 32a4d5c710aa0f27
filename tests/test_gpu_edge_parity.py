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


def check(ctx, oracle, idx, flags, seqs, lens, fresh=0):
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
                                   ["-P", "-n", "0"], ["-P", "-n", "3"], ["-P", "-S", "-n", "2"],
                                   # penalties beyond the 12-score LDS window of bucket states: the global row is used as well
                                   ["-n", "3", "-O", "20", "-E", "13", "-e", "2"], ["-n", "4", "-M", "14", "-O", "15", "-E", "2", "-o", "2", "-e", "3"]])
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
    """Reads shorter than the seed (no longer read before them: D_seed is the calloc'd zeros), at the seed length, and 250-base reads."""
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
    monkeypatch.setenv("BWB_POOL_GB", "0")  # clamps to the 256 MB floor: 262 144 chunks, not enough for 5 000 reads at once
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "o.fq"), 5000, 100, 8, sub=2.5)
    st = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    assert st.n_overflow_reads > 0
    # penalties above 63 (32-byte entries, LHeap::far: buckets beyond the 64-bucket window) through the same re-run classes
    st = check(ctx, oracle, idx, ["-n", "3", "-M", "80", "-O", "90", "-E", "70"], seqs[:3000], lens[:3000])
    ctx.close()


@pytest.mark.parametrize("env", [{"BWB_FORCE_SLICES": "1"}, {"BWB_SLICE_ITERS": "150"}, {"BWB_SLICE_ITERS": "40", "BWB_FORCE_POS64": "1"},
                                 {"BWB_SLICE_ITERS": "300", "BWB_POOL_GB": "0"}])
def test_parked_and_resumed_reads_are_exact(mid, oracle, monkeypatch, env):
    """Slices: a launch parks the reads under way (all per-lane state to the save area) and the next launch resumes them.
    BWB_FORCE_SLICES parks when the cursor runs out; BWB_SLICE_ITERS parks every wave after that many loop iterations, so
    every read is parked and resumed many times, in every mode (pop, exact tail, -P seeding), with a starved pool too."""
    d, fa = mid
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "h.fq"), 4000, 100, 9, sub=2.0, indel=1.0, npct=1.0)
    st = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    assert st.n_parked_reads > 0 and st.launches_search > 1
    check(ctx, oracle, idx, ["-n", "4", "-o", "2", "-e", "3"], seqs[:1500], lens[:1500])
    check(ctx, oracle, idx, ["-P", "-n", "2"], seqs, lens)
    check(ctx, oracle, idx, ["-S", "-n", "2"], seqs, lens)
    check(ctx, oracle, idx, ["-n", "3", "-O", "20", "-E", "13", "-e", "2"], seqs[:1500], lens[:1500])
    # more than 255 heap buckets: best_score has its own word in the save area (round 3 packed it into 8 bits: ADVICE r3), and
    # entry scores above 255 (the 8-bit wrap of aln_entry_t.score at inexact_match.c:309; 16-bit scores in the hit records)
    check(ctx, oracle, idx, ["-n", "3", "-M", "40", "-O", "45", "-E", "10"], seqs[:1500], lens[:1500])
    check(ctx, oracle, idx, ["-n", "5", "-M", "52", "-O", "60", "-E", "30"], seqs[:600], lens[:600])
    # penalties above 63: LHeap::far's window top-up, side_flush and the save / restore of neW, cb, cst across slices
    check(ctx, oracle, idx, ["-n", "3", "-M", "80", "-O", "90", "-E", "70"], seqs[:1500], lens[:1500])
    ctx.close()


def test_streamed_slots_match_oracle(mid_ctx, oracle):
    """The streaming interface: batches of different sizes go through the slots back to back (every slice parks its
    unfinished reads for the next), slots are reused, and each batch's bytes equal the oracle's."""
    d, fa, ctx, idx = mid_ctx
    seqs, lens = synth_reads(fa, str(d / "s.fq"), 9000, 100, 21, sub=2.0, indel=1.0, npct=1.0)
    flags = ["-n", "3"]
    p = bw.params(flags)
    cuts = [0, 2500, 2600, 5600, 5600, 7000, 9000]   # includes an empty batch and a 100-read one
    want = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        data, _, _ = oracle.align_encoded(idx, seqs[lo:hi], lens[lo:hi], oracle.params(flags))
        want.append(data)
    ctx.flush()
    ctx.reset_stats()
    got = [None] * len(want)
    nslot = bw.MAX_SLOTS
    for j, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        slot = j % nslot
        if j >= nslot:   # the slot is reused: take its result first
            off, alns = ctx.slot_result(slot)
            got[j - nslot] = bw.aln_bytes(off, alns)
        ctx.slot_upload(slot, p, seqs[lo:hi], lens[lo:hi])
        ctx.slot_submit(slot)
    for j in range(max(0, len(want) - nslot), len(want)):
        off, alns = ctx.slot_result(j % nslot)
        got[j] = bw.aln_bytes(off, alns)
    ctx.flush()
    assert got == want
    st = ctx.stats()
    _, ost, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags))
    assert st.visits_single + st.visits_alphabet == ost.visits_single + ost.visits_alphabet
    assert st.heap_pops == ost.heap_pops and st.heap_pushes == ost.heap_pushes
    assert st.n_parked_reads > 0
    # a different parameter set while slots are alive flushes first; then the one-batch interface still works
    ctx.slot_upload(0, p, seqs[:500], lens[:500])
    ctx.slot_submit(0)
    check(ctx, oracle, idx, ["-n", "2"], seqs[:800], lens[:800])


@pytest.mark.parametrize("env", [{"BWB_CALCD_AHEAD": "2"}, {"BWB_CALCD_AHEAD": "1", "BWB_SLICE_ITERS": "200"}, {"BWB_CALCD_AHEAD": "3", "BWB_CALCD_PRIO": "1", "BWB_POOL_GB": "0"}])
def test_calc_d_queued_ahead_on_a_second_stream_is_exact(mid, oracle, monkeypatch, env):
    """BWB_CALCD_AHEAD: kl_calc_d of the batches uploaded beyond the submitted one runs on a second kernel stream beside the search
    slices (the search of a batch waits for its own kl_calc_d through an event).  Batches uploaded several slots ahead, slots reused,
    short reads that inherit D_seed across batch heads, an empty batch: same bytes and work counters as the oracle."""
    d, fa = mid
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "ahead.fq"), 12000, 100, 33, sub=2.0, indel=1.0, npct=1.0)
    lens = lens.copy()
    lens[::7] = 30  # reads not longer than the seed: their D_seed is inherited (k_dseed_inherit runs behind kl_calc_d on the second stream)
    flags = ["-n", "3"]
    p = bw.params(flags)
    cuts = list(range(0, 12001, 1000))
    cuts.insert(5, cuts[5])  # an empty batch
    want = [oracle.align_encoded(idx, seqs[lo:hi], lens[lo:hi], oracle.params(flags))[0] for lo, hi in zip(cuts[:-1], cuts[1:])]
    nb, nslot = len(want), bw.MAX_SLOTS
    got = [None] * nb
    # uploads run up to four batches ahead of the submits
    up = sub = 0
    def upload(j):
        slot = j % nslot
        if j >= nslot:
            off, alns = ctx.slot_result(slot)
            got[j - nslot] = bw.aln_bytes(off, alns)
        ctx.slot_upload(slot, p, seqs[cuts[j]:cuts[j + 1]], lens[cuts[j]:cuts[j + 1]])
    while sub < nb:
        while up < nb and up < sub + 4:
            upload(up); up += 1
        ctx.slot_submit(sub % nslot); sub += 1
    for j in range(max(0, nb - nslot), nb):
        off, alns = ctx.slot_result(j % nslot)
        got[j] = bw.aln_bytes(off, alns)
    ctx.flush()
    assert got == want
    st = ctx.stats()
    assert st.launches_calc_d >= nb - 1
    check(ctx, oracle, idx, ["-n", "2"], seqs[:800], lens[:800])  # the one-batch interface on the same context
    ctx.close()


@pytest.mark.parametrize("env", [{"BWB_DTAB": "1", "BWB_DTAB_K": "8"}, {"BWB_DTAB": "1", "BWB_DTAB_K": "12", "BWB_POOL_GB": "3"},
                                 {"BWB_DTAB": "1", "BWB_DTAB_K": "5", "BWB_FORCE_POS64": "1", "BWB_CALCD_AHEAD": "2"}])
def test_calculate_d_table_is_exact(mid, oracle, monkeypatch, env):
    """The calculate_d table (bwb_lane.h: DTab): kl_calc_d starts a read - and its seed - from the state after the first K steps, looked up by
    the last K bases.  Same bytes, same rank-visit / pop / push counters as the oracle (any difference in a D byte changes the pruning), for the
    multi-genome alphabet, -S (its own table), -P, ragged lengths around K, reads with N among the last K bases; and the table really is
    used: fewer buckets fetched by kl_calc_d than without it, the same visits."""
    d, fa = mid
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    seqs, lens = synth_reads(fa, str(d / "dtab.fq"), 6000, 100, 61, sub=2.0, indel=1.0, npct=1.0)
    lens = lens.copy()
    K = int(env["BWB_DTAB_K"])
    for j, ln in enumerate([K - 1, K, K + 1, K + 2, 13, 32, 33, 44, 45, 57]):   # around K, around the seed length (32: its own phase), 45 = 32 + 13
        lens[j::97] = max(1, ln)
    seqs = seqs.copy()
    seqs[5::53, 3] = 4                      # an N among the read's first bases (the seed's table lookup sees the bases 20..31)
    for i in range(7, len(lens), 41):
        seqs[i, max(0, int(lens[i]) - 4)] = 4  # an N among the last K bases: the read takes the ordinary path, its seed may not
    st = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    table_buckets, table_visits = st.bucket_loads_calc_d, st.visits_calc_d
    check(ctx, oracle, idx, ["-n", "0"], seqs[:2000], lens[:2000])
    check(ctx, oracle, idx, ["-S", "-n", "2"], seqs[:3000], lens[:3000])       # another alphabet: the table is rebuilt
    sel = np.where(lens[:3000] >= 13)[0]                                       # (-P: reads shorter than 12 are not representable, align.c:174-186)
    check(ctx, oracle, idx, ["-P", "-n", "2"], seqs[sel], lens[sel])           # back to the multi-genome one
    check(ctx, oracle, idx, ["-n", "3", "-l", "20", "-k", "1"], seqs[:3000], lens[:3000])
    check(ctx, oracle, idx, ["-n", "2", "-l", "0"], seqs[:2000], lens[:2000])  # no seed
    ctx.close()
    monkeypatch.setenv("BWB_DTAB", "0")
    ctx = bw.Context(fa + ".bwt")
    st0 = check(ctx, oracle, idx, ["-n", "3"], seqs, lens)
    ctx.close()
    assert st0.visits_calc_d == table_visits
    assert table_buckets < st0.bucket_loads_calc_d, (table_buckets, st0.bucket_loads_calc_d)


def test_streamed_results_are_published_before_the_host_reads_them(mid_ctx):
    """Long slices: a slot's parked reads finish INSIDE the next slot's slice, and the host fetches status, counts, offsets and
    the hit log on another stream while that kernel is still running.  The kernel publishes a read's results (release fence)
    before it moves the slot's counter (publish_done, bwb_lane.h); here every chunk of a 2 M-read stream must equal what the
    one-batch interface - which reads back only after every kernel has ended - gives for the same reads."""
    d, fa, ctx, idx = mid_ctx
    B, nbatch = 250000, 8
    seqs, lens = synth_reads(fa, str(d / "pub.fq"), B * nbatch, 100, 33, sub=1.5, indel=0.5, npct=0.5)
    p = bw.params(["-n", "3"])
    want = []
    for j in range(nbatch):
        off, alns = ctx.align(p, seqs[j * B:(j + 1) * B], lens[j * B:(j + 1) * B])
        want.append((off.tobytes(), alns.tobytes()))  # (raw records: serialising 2 M reads in Python would take minutes)
    ctx.flush()
    ctx.reset_stats()
    got = [None] * nbatch
    nslot = 3
    for j in range(nbatch):
        slot = j % nslot
        if j >= nslot:
            off, alns = ctx.slot_result(slot)
            got[j - nslot] = (off.tobytes(), alns.tobytes())
        ctx.slot_upload(slot, p, seqs[j * B:(j + 1) * B], lens[j * B:(j + 1) * B])
        ctx.slot_submit(slot)
    for j in range(nbatch - nslot, nbatch):
        off, alns = ctx.slot_result(j % nslot)
        got[j] = (off.tobytes(), alns.tobytes())
    ctx.flush()
    assert [len(g[1]) for g in got] == [len(w[1]) for w in want]
    assert got == want
    assert ctx.stats().n_parked_reads > 0


def test_unrepresentable_reads_get_empty_records(mid_ctx, oracle):
    """A read longer than 255 bases (aln_entry_t.i is 8-bit, align.h:104) no longer fails the batch: empty record, the others
    exact.  (An EMPTY read is different: like in the reference its root entry is a hit with the whole index as interval.)"""
    d, fa, ctx, idx = mid_ctx
    seqs, lens = synth_reads(fa, str(d / "u.fq"), 300, 100, 33)
    empty = (0).to_bytes(4, "little")

    def want_with_hole(flags, hole):
        a, _, _ = oracle.align_encoded(idx, seqs[:hole], lens[:hole], oracle.params(flags))
        b, _, _ = oracle.align_encoded(idx, seqs[hole + 1:], lens[hole + 1:], oracle.params(flags))
        return a + empty + b

    big = np.full((len(lens), 300), 4, dtype=np.uint8)
    big[:, :100] = seqs
    lens2 = lens.copy()
    lens2[7] = 300
    big[7, :] = 0
    off, alns = ctx.align(bw.params(["-n", "2"]), big, lens2)
    assert bw.aln_bytes(off, alns) == want_with_hole(["-n", "2"], 7)
    # -P with a read shorter than 12 bases: the same
    lens4 = lens.copy()
    lens4[3] = 11
    off, alns = ctx.align(bw.params(["-P", "-n", "1"]), seqs, lens4)
    assert bw.aln_bytes(off, alns) == want_with_hole(["-P", "-n", "1"], 3)
    # an empty read, and one of a single base: whatever the reference does with them (here: the oracle)
    lens5 = lens.copy()
    lens5[5] = 0
    lens5[9] = 1
    check(ctx, oracle, idx, ["-n", "2"], seqs, lens5)


def test_rank_bench_layouts_agree(mid_ctx):
    """The two rank micro-benchmarks (octet-cooperative, one query per lane) answer the same queries: same checksum."""
    d, fa, ctx, idx = mid_ctx
    _, a = ctx.rank_bench(1 << 16, iters=1, seed=5)
    _, b = ctx.rank_bench(1 << 16, iters=1, seed=5, lane=True)
    assert a == b and a != 0


def test_bucket_load_counter(mid_ctx, oracle):
    """device buckets fetched <= algorithmic visits (a same-bucket L-1/U pair is fetched once), and not zero"""
    d, fa, ctx, idx = mid_ctx
    seqs, lens = synth_reads(fa, str(d / "b.fq"), 2000, 100, 44)
    st = check(ctx, oracle, idx, ["-n", "2"], seqs, lens)
    vis_search = st.visits_single + st.visits_alphabet - st.visits_calc_d
    assert 0 < st.bucket_loads_calc_d <= st.visits_calc_d
    assert 0 < st.bucket_loads_search <= vis_search + 2 * st.heap_pops  # (visits of pruned / unfinished work are not counted as algorithmic)


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


def fuzz_case(rng, d, fa, tag):
    """one case of tools/fuzz_parity.py's generator: flags, reads of one or two interleaved lengths (short reads inherit D_seed)"""
    ln = rng.choice([24, 36, 50, 76, 100, 100, 125, 150, 200])
    seqs, lens = synth_reads(fa, str(d / f"fz{tag}.fq"), rng.choice([300, 800, 1500]), ln, rng.randrange(10000),
                             sub=rng.choice([0.5, 1.0, 2.0, 4.0]), indel=rng.choice([0.0, 1.0, 5.0]), npct=rng.choice([0.0, 1.0, 10.0]))
    if rng.random() < 0.5:
        s2, l2 = synth_reads(fa, str(d / f"fz{tag}b.fq"), rng.choice([100, 400]), rng.choice([20, 28, 33, 64, 100]), rng.randrange(10000), sub=2.0, indel=2.0, npct=2.0)
        w = max(seqs.shape[1], s2.shape[1])
        allseq = np.full((len(lens) + len(l2), w), 4, dtype=np.uint8)
        allseq[:len(lens), :seqs.shape[1]] = seqs
        allseq[len(lens):, :s2.shape[1]] = s2
        order = np.random.default_rng(rng.randrange(1 << 30)).permutation(len(allseq))
        seqs, lens = allseq[order], np.concatenate([lens, l2])[order]
    flags = ["-n", str(rng.choice([0, 1, 2, 3, 3, 4])), "-o", str(rng.choice([0, 1, 1, 2, 3])), "-e", str(rng.choice([0, 2, 6])),
             "-l", str(rng.choice([0, 16, 32, 32, 60])), "-k", str(rng.choice([0, 1, 2, 3])), "-M", str(rng.choice([1, 3, 3, 5])),
             "-O", str(rng.choice([3, 11, 11])), "-E", str(rng.choice([1, 4, 4])), "-m", str(rng.choice([200, 5000, 3000000]))]
    if rng.random() < 0.2:
        flags.append("-S")
    if rng.random() < 0.2:
        flags.append("-P")
    return flags, seqs, lens


@pytest.mark.parametrize("env,seed,cases", [({}, 5001, 28), ({"BWB_SLICE_ITERS": "90"}, 5002, 12), ({"BWB_DTAB": "1", "BWB_DTAB_K": "7"}, 5003, 16)])
def test_fuzz_sweep_of_parameters_read_lengths_and_error_rates(mid, oracle, monkeypatch, env, seed, cases):
    """40 seeded cases of tools/fuzz_parity.py's generator inside the GPU suite: random -n/-o/-e/-l/-k/-M/-O/-E/-m, -S, -P,
    read lengths 24..200 with a second length interleaved, substitution / indel / N rates - 28 cases in one launch each, 12 with every
    wave parked after 90 loop iterations (every read parked and resumed dozens of times), 16 (round 6) with the calculate_d table forced
    (K = 7; rebuilt whenever a case switches the alphabet).  Bytes, visits, pops and pushes equal to the oracle's (serial reference:
    fresh_dseed = 0)."""
    import random
    d, fa = mid
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = bw.Context(fa + ".bwt")
    idx = oracle.load_index(fa + ".bwt")
    rng = random.Random(seed)
    done = 0
    try:
        for c in range(cases):
            flags, seqs, lens = fuzz_case(rng, d, fa, f"{seed}_{c}")
            try:
                bw.params(flags)
                check(ctx, oracle, idx, flags, seqs, lens)
                done += 1
            except bw.BwbError as e:  # a refused parameter combination is loud, not wrong
                assert "supported" in str(e) or "must be" in str(e), (flags, str(e))
    finally:
        ctx.close()
    assert done >= cases - 2


def test_short_reads_inherit_the_last_longer_reads_dseed(mid_ctx, oracle):
    """Reads <= seed_length mixed with longer ones: D_seed of a short read = that of the last longer read before it (the serial
    reference's single buffer, inexact_match.c:35,62-65); in one batch, and streamed in small batches with the carried read."""
    d, fa, ctx, idx = mid_ctx
    parts = [synth_reads(fa, str(d / f"sh{k}.fq"), cnt, ln, 70 + k, sub=3.0, indel=4.0, npct=4.0)
             for k, (ln, cnt) in enumerate([(25, 30), (100, 40), (30, 50), (32, 30), (64, 20), (20, 60), (33, 10), (150, 12), (31, 80)])]
    stride = max(p[0].shape[1] for p in parts)
    seqs = np.full((sum(len(p[1]) for p in parts), stride), 4, dtype=np.uint8)
    lens = np.concatenate([p[1] for p in parts])
    o = 0
    for s_, l_ in parts:
        seqs[o:o + len(l_), :s_.shape[1]] = s_
        o += len(l_)
    order = np.random.default_rng(3).permutation(len(lens))  # interleave the lengths
    seqs, lens = seqs[order], lens[order]
    for flags in (["-n", "2"], ["-n", "3", "-k", "1"], ["-P", "-n", "2"], ["-n", "2", "-l", "64"]):
        check(ctx, oracle, idx, flags, seqs, lens)
        fresh, _, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags), fresh_dseed=1)
        stale, _, _ = oracle.align_encoded(idx, seqs, lens, oracle.params(flags), fresh_dseed=0)
        # streamed in batches of 23 reads: the source of a batch's leading short reads lies in an earlier batch (the carry)
        p = bw.params(flags)
        sl = p.seed_length
        got, last = b"", None
        ctx.flush()
        for lo in range(0, len(lens), 23):
            hi = min(lo + 23, len(lens))
            ctx.slot_upload(0, p, seqs[lo:hi], lens[lo:hi], carry=last)
            ctx.slot_submit(0)
            off, alns = ctx.slot_result(0)
            got += bw.aln_bytes(off, alns)
            for i in range(lo, hi):
                ok = lens[i] > sl
                if ok and p.use_precalc:
                    ok = not (seqs[i, :12] > 3).any()
                if ok:
                    last = seqs[i, :lens[i]].copy()
        assert got == stale
    assert fresh != stale  # (the fixture discriminates)
