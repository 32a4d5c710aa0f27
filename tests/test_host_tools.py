"""Host-side C tools (bwbble_amd/host): index builder, .aln reader/writer, FASTQ reader.
CPU tests check file-format parity with the reference's golden files; GPU tests run the CLI end to end."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import bwbble_amd as bw
from golden.make_golden import ALIGN_CONFIGS


def run(cmd):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout


@pytest.fixture(scope="module")
def toy_dir(built, golden, tmp_path_factory):
    d = tmp_path_factory.mktemp("toy")
    shutil.copy(os.path.join(golden, "toy.fa"), d / "toy.fa")
    run([bw.HOST_BIN, "index", str(d / "toy.fa")])
    return d


def test_index_files_identical_to_reference(toy_dir, golden):
    """`bwbble index` (own suffix sort) writes the reference's .bwt and .ann byte for byte (bwt.c:66-82, io.c:292-296)."""
    for ext in (".bwt", ".ann"):
        assert open(toy_dir / ("toy.fa" + ext), "rb").read() == open(os.path.join(golden, "toy.fa" + ext), "rb").read()
    ref = np.fromfile(toy_dir / "toy.fa.ref", dtype=np.uint8)
    n = len(ref) // 2
    compl = np.array([0, 15, 8, 7, 4, 11, 12, 3, 2, 13, 10, 5, 6, 9, 14, 1], dtype=np.uint8)  # iupacCompl io.h:32
    assert np.array_equal(ref[n:], compl[ref[:n]][::-1])


def test_index_edge_cases(built, tmp_path):
    """Lower-case input, non-IUPAC characters (-> N), repeats/duplicated records, an odd number of characters:
    checked against a naive suffix sort."""
    fa = tmp_path / "e.fa"
    fa.write_text(">a\nacgtACGTnnRYKM\nACGTACGTACGTA\n>b dup\nACGTACGTACGTA\n>c\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAX\n")
    run([bw.HOST_BIN, "index", str(fa)])
    b = bw.BwtFile(str(fa) + ".bwt", load_sa=True)
    text = np.fromfile(str(fa) + ".ref", dtype=np.uint8)
    n = len(text)
    assert b.length == n + 1
    sa = sorted(range(n + 1), key=lambda i: bytes(text[i:]))  # end of text smallest, like is_sa (is.c:197-206)
    bwt = [0 if s == 0 else int(text[s - 1]) for s in sa]
    got = [(int(b.bwt[i >> 3]) >> (28 - 4 * (i & 7))) & 15 for i in range(n + 1)]
    assert got == bwt
    assert b.sa0_index == sa.index(0)
    assert [int(v) for v in b.SA] == [sa[i] for i in range(0, n + 1, 32)]
    cnt = np.bincount(text, minlength=16)
    assert [int(v) for v in b.C] == [0] + list(np.cumsum(cnt))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n0", "n3", "n4gap", "s2", "p2"])
def test_cli_align_writes_reference_aln(toy_dir, golden, name):
    out = toy_dir / f"{name}.aln"
    log = run([bw.HOST_BIN, "align"] + ALIGN_CONFIGS[name] + [str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(out)])
    assert "Processed 600 reads" in log
    assert open(out, "rb").read() == open(os.path.join(golden, f"toy_{name}.aln"), "rb").read()


@pytest.mark.gpu
def test_cli_align_P_leaves_the_reference_pre_table(toy_dir, golden):
    """The first `align -P` on an index writes <fasta>.pre like the reference does (align.c:59-65, 200-224): 16.7 M interval lists,
    74 MB for the toy index - byte-identical to the reference's table (its SHA-256 is the committed fixture; host/precalc.c)."""
    import hashlib
    pre = toy_dir / "toy.fa.pre"
    if pre.exists():
        pre.unlink()
    log = run([bw.HOST_BIN, "align", "-P", "-n", "0", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(toy_dir / "pre_p0.aln")])
    assert "Pre-calculating SA intervals" in log and pre.exists()
    h = hashlib.sha256()
    with open(pre, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    assert h.hexdigest() == open(os.path.join(golden, "toy.fa.pre.sha256")).read().split()[0]
    assert open(toy_dir / "pre_p0.aln", "rb").read() == open(os.path.join(golden, "toy_p0.aln"), "rb").read()
    log = run([bw.HOST_BIN, "align", "-P", "-n", "0", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(toy_dir / "pre_p0b.aln")])
    assert "Pre-calculating" not in log  # an existing table is left alone


@pytest.mark.gpu
def test_cli_align_multi_chunk_order(toy_dir, golden):
    """Chunks pulled by the per-GPU host threads are written back in input order."""
    out = toy_dir / "chunks.aln"
    env = dict(os.environ, BWB_CHUNK="97")
    subprocess.run([bw.HOST_BIN, "align", "-n", "3", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(out)],
                   check=True, env=env, stdout=subprocess.DEVNULL)
    assert open(out, "rb").read() == open(os.path.join(golden, "toy_n3.aln"), "rb").read()


@pytest.mark.gpu
def test_cli_align_two_workers_on_one_device(toy_dir, golden):
    """`align -g 2` with both workers mapped to device 0: two host threads, two contexts, shared chunk cursor, ordered writer
    (the product's multi-GPU path, host/align_gpu.c) - the bytes must not depend on which worker took which chunk."""
    out = toy_dir / "g2.aln"
    env = dict(os.environ, BWB_DEVICE_MAP="0,0", BWB_CHUNK="97", BWB_POOL_GB="1")
    log = subprocess.run([bw.HOST_BIN, "align", "-n", "3", "-g", "2", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(out)],
                         check=True, env=env, stdout=subprocess.PIPE, text=True).stdout
    assert "GPUs: 2" in log
    assert open(out, "rb").read() == open(os.path.join(golden, "toy_n3.aln"), "rb").read()


@pytest.mark.gpu
def test_cli_align_four_workers_on_one_device(toy_dir, golden):
    """`align -g 4` (the 8-GPU path of config C4 with as many workers as one device reasonably takes): four host threads and
    contexts on device 0, 7 chunks, ordered writer - the reference's bytes."""
    out = toy_dir / "g4.aln"
    env = dict(os.environ, BWB_DEVICE_MAP="0,0,0,0", BWB_CHUNK="97", BWB_POOL_GB="1", BWB_BLOCKS_PER_CU="1", BWB_CALCD_BLOCKS_PER_CU="1")
    log = subprocess.run([bw.HOST_BIN, "align", "-n", "3", "-g", "4", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(out)],
                         check=True, env=env, stdout=subprocess.PIPE, text=True).stdout
    assert "GPUs: 4" in log
    assert open(out, "rb").read() == open(os.path.join(golden, "toy_n3.aln"), "rb").read()


@pytest.mark.gpu
def test_cli_more_gpus_than_present_fails_loudly(toy_dir, golden):
    n = bw.device_count()
    r = subprocess.run([bw.HOST_BIN, "align", "-g", str(n + 1), str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(toy_dir / "y.aln")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "asked for" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("fq,aln,sam", [("toy.fq", "toy_n3.aln", "toy_n3.sam"), ("ragged.fq", "ragged_n4gap.aln", "ragged_n4gap.sam")])
def test_cli_aln2sam_writes_reference_sam(toy_dir, golden, fq, aln, sam):
    out = toy_dir / (sam + ".out")
    run([bw.HOST_BIN, "aln2sam", str(toy_dir / "toy.fa"), os.path.join(golden, fq), os.path.join(golden, aln), str(out)])
    assert open(out).read() == open(os.path.join(golden, sam)).read()


@pytest.mark.gpu
def test_locate_matches_oracle(toy_dir, oracle, golden):
    b = bw.BwtFile(os.path.join(golden, "toy.fa.bwt"), load_sa=True)
    ctx = bw.Context(b)
    ctx.set_sa(b.SA)
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.integers(0, b.length, 4000), [0, b.length - 1, b.sa0_index, 32, 31]]).astype(np.uint64)
    got = ctx.locate(rows)
    idx = oracle.load_index(os.path.join(golden, "toy.fa.bwt"), load_sa=True)
    want = np.array([oracle.lib.bwb_or_SA(idx, int(r)) for r in rows], dtype=np.uint64)
    assert np.array_equal(got, want)
    ctx.close()


def test_cli_align_without_gpu_fails_loudly(toy_dir, golden):
    if bw.device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([bw.HOST_BIN, "align", str(toy_dir / "toy.fa"), os.path.join(golden, "toy.fq"), str(toy_dir / "x.aln")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stdout


def _repeat_rich_fasta(path):
    """Deterministic FASTA with what real assemblies have and iid text has not: a long N run, a homopolymer, a tandem array, an
    exact segmental duplication (also across records), on top of random sequence."""
    rng = np.random.default_rng(20240917)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: acgt[rng.integers(0, 4, n)].tobytes()
    dup = rnd(40000)
    rec1 = rnd(50000) + b"N" * 300000 + rnd(30000) + b"A" * 120000 + rnd(20000) + dup + rnd(10000) + b"ACGGT" * 16000 + rnd(5000)
    rec2 = rnd(7000) + dup + b"N" * 70000 + rnd(9000) + b"T" * 30000 + rnd(3000)
    with open(path, "wb") as f:
        for name, rec in ((b"chrA", rec1), (b"chrB with spaces", rec2)):
            f.write(b">" + name + b"\n")
            for i in range(0, len(rec), 60):
                f.write(rec[i:i + 60] + b"\n")


# sha256 of the .bwt the REAL reference (oracle/_ref/bwbble index, sais-lite) writes for _repeat_rich_fasta
REPEAT_RICH_BWT_SHA256 = "92fab3456babf506484605f2789d141d69ea37febc683174da300278151b8dea"


def test_index_long_repeats_match_reference(built, tmp_path):
    """N runs / homopolymers / tandem arrays / exact duplications: the key-refinement suffix sort hands them to its prefix
    doubling phase; the .bwt must still be the reference's, byte for byte (index.c not scale-safe on real genomes)."""
    import hashlib
    fa = tmp_path / "rep.fa"
    _repeat_rich_fasta(fa)
    log = run([bw.HOST_BIN, "index", str(fa)])
    assert "prefix doubling" in log
    data = open(str(fa) + ".bwt", "rb").read()
    # (1) the BWT inverts to the text: walk LF from the row of the empty suffix (row 0), is.c:197-243 row convention
    b = bw.BwtFile(str(fa) + ".bwt")
    text = np.fromfile(str(fa) + ".ref", dtype=np.uint8)
    n = len(text)
    assert b.length == n + 1
    w = np.asarray(b.bwt, dtype=np.uint32)
    L = ((w[:, None] >> (28 - 4 * np.arange(8, dtype=np.uint32))[None, :]) & 15).astype(np.int16).reshape(-1)[:n + 1] + 1
    L[b.sa0_index] = 0  # the row whose suffix is the whole text: its "previous character" is the end-of-text sentinel
    order = np.argsort(L, kind="stable")
    LF = np.empty(n + 1, dtype=np.int64)
    LF[order] = np.arange(n + 1)
    out = np.empty(n, dtype=np.uint8)
    i = 0
    Ll, LFl = (L - 1).tolist(), LF.tolist()
    for k in range(n - 1, -1, -1):
        out[k] = Ll[i]
        i = LFl[i]
    assert i == b.sa0_index and np.array_equal(out, text)
    # (2) byte-identical to the reference's file (hash recorded from the reference run in the build container) ...
    assert hashlib.sha256(data).hexdigest() == REPEAT_RICH_BWT_SHA256
    # ... and, where the reference binary is present, re-checked against a fresh run of it
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "bwbble")
    if os.path.exists(ref_bin) and os.path.isdir("/root/reference"):
        fb = tmp_path / "ref.fa"
        shutil.copy(fa, fb)
        run([ref_bin, "index", str(fb)])
        assert open(str(fb) + ".bwt", "rb").read() == data
        assert open(str(fb) + ".ann", "rb").read() == open(str(fa) + ".ann", "rb").read()


@pytest.mark.gpu
def test_cli_wgsim_shaped_fastq(toy_dir, golden):
    """FASTQ in the shape of the reference's test_data/sim_chr21_N100.fastq (wgsim names, bare '+' lines) plus '+name' lines,
    lower-case bases, names with blanks, blank lines and no final newline: host/reads.c against the reference's fastq2reads
    (io.c:410-515) through align and aln2sam - names, sequences and qualities all end up in the SAM text."""
    aln, sam = toy_dir / "w.aln", toy_dir / "w.sam"
    fq = os.path.join(golden, "wgsim100.fq")
    log = run([bw.HOST_BIN, "align", "-n", "2", str(toy_dir / "toy.fa"), fq, str(aln)])
    assert "Processed 100 reads" in log
    assert open(aln, "rb").read() == open(os.path.join(golden, "wgsim100_n2.aln"), "rb").read()
    run([bw.HOST_BIN, "aln2sam", str(toy_dir / "toy.fa"), fq, str(aln), str(sam)])
    assert open(sam).read() == open(os.path.join(golden, "wgsim100_n2.sam")).read()


@pytest.mark.gpu
def test_cli_on_the_references_own_test_fastq(toy_dir, golden):
    """the reference's test_data/sim_chr21_N100.fastq itself (config C1's literal input: 100 wgsim reads of chr21) through `align` with the
    CLI default -n 0 and with -n 2, and through `aln2sam`, on the toy index: the reference's .aln and .sam bytes (real chr21 reads do not
    map to the synthetic text - nearly every record is empty -, so this pins the parser, the record writer and the unmapped-read SAM lines)"""
    fq = os.path.join(golden, "sim_chr21_N100.fastq")
    for name, flags in (("n0", []), ("n2", ["-n", "2"])):  # (no flag at all = the reference's default, -n 0: align.c:26)
        aln = toy_dir / f"chr21_{name}.aln"
        log = run([bw.HOST_BIN, "align"] + flags + [str(toy_dir / "toy.fa"), fq, str(aln)])
        assert "Processed 100 reads" in log
        assert open(aln, "rb").read() == open(os.path.join(golden, f"sim_chr21_N100_{name}.aln"), "rb").read()
    sam = toy_dir / "chr21_n2.sam"
    run([bw.HOST_BIN, "aln2sam", str(toy_dir / "toy.fa"), fq, str(toy_dir / "chr21_n2.aln"), str(sam)])
    assert open(sam).read() == open(os.path.join(golden, "sim_chr21_N100_n2.sam")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["1000000", "7"])
def test_cli_short_reads_match_serial_reference(toy_dir, golden, chunk):
    """the CLI on short.fq, in one chunk and in chunks of 7 reads (the D_seed source of a chunk's first short reads is carried over)"""
    out = toy_dir / f"short{chunk}.aln"
    env = dict(os.environ, BWB_CHUNK=chunk)
    subprocess.run([bw.HOST_BIN, "align", "-n", "3", "-k", "1", str(toy_dir / "toy.fa"), os.path.join(golden, "short.fq"), str(out)],
                   check=True, env=env, stdout=subprocess.DEVNULL)
    assert open(out, "rb").read() == open(os.path.join(golden, "short_n3k1_t1.aln"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_shard_one_fastq(built, tmp_path, ranks):
    """bench.py --gpus N launches its own N ranks (here all on GPU 0, BWB_BENCH_SHARE_DEVICE): the index is replicated, rank r
    aligns shard r of the logical FASTQ, every rank re-aligns a sample of its neighbour's shard and the checksums must agree;
    the line carries every rank's own rate (the 8-GPU protocol of config C4, as far as one device can exercise it)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BWB_BENCH_SHARE_DEVICE="1", BWB_POOL_GB="2", BWB_BENCH_DIR=str(tmp_path), BWB_BLOCKS_PER_CU="1", BWB_CALCD_BLOCKS_PER_CU="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--genome-mb", "2", "--pool", "24000", "--reads", "6000",
                        "--steps", "3", "--warmup", "1", "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line["shard_sample_parity"] is True and line["config"]["reads_per_gpu_per_step"] == 6000 and line["value"] > 0
    assert 0 < line["per_rank_reads_per_s"]["min"] <= line["per_rank_reads_per_s"]["max"]
    assert line["roofline"]["frac"] <= 1.0 and all(k["device_frac"] <= 1.0 for k in line["roofline"]["kernels"].values())
    if ranks != 2:
        return
    # asking for more ranks than the launcher started is an error, not a silent one-GPU run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--no-extras"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "launcher started" in bad.stderr


def test_index_from_external_suffix_array(built, golden, tmp_path):
    """`index -e <sa>` (esa2bwt, bwt.c:132-158): the BWT from a 40-bit external suffix array and the .ref file, streamed.
    The same .bwt as the built-in sorter's, which is the reference's (golden toy.fa.bwt)."""
    fa = tmp_path / "toy.fa"
    shutil.copy(os.path.join(golden, "toy.fa"), fa)
    sa = tmp_path / "toy.sa5"
    subprocess.run([bw.HOST_BIN, "index", str(fa)], check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, BWB_DUMP_SA=str(sa)))
    want = open(str(fa) + ".bwt", "rb").read()
    assert want == open(os.path.join(golden, "toy.fa.bwt"), "rb").read()
    assert os.path.getsize(sa) == 5 * (bw.BwtFile(str(fa) + ".bwt").length - 1)
    os.remove(str(fa) + ".bwt")
    run([bw.HOST_BIN, "index", "-e", str(sa), str(fa)])
    assert open(str(fa) + ".bwt", "rb").read() == want
    # (no cross-check with the reference here: its esa2bwt reads each 5-byte entry into an uninitialised 8-byte variable,
    #  bwt.c:143-145, and the binary built from it crashes on this input; the .bwt is pinned to the reference's through the
    #  built-in path two lines up)
    bad = subprocess.run([bw.HOST_BIN, "index", "-e", str(tmp_path / "missing.sa5"), str(fa)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert bad.returncode != 0 and "Cannot open the ext SA file" in bad.stdout


def _fastq2reads_model(path):
    """fastq2reads (mg-aligner/io.c:410-515) restated record by record: skip to '@', name = rest of the line (256 chars at most),
    sequence line -> codes (A0 G1 C2 T3, anything else 4), skip to '+', skip its line, quality line."""
    data = open(path, "rb").read().decode("latin-1")
    code = {"A": "0", "a": "0", "G": "1", "g": "1", "C": "2", "c": "2", "T": "3", "t": "3"}
    out, p, n = [], 0, len(data)
    while True:
        p = data.find("@", p)
        if p < 0:
            break
        e = data.find("\n", p)
        name = data[p + 1:e][:256]
        s0 = e + 1
        e = data.find("\n", s0)
        seq = data[s0:e]
        p = data.find("+", e)
        e = data.find("\n", p)
        q0 = e + 1
        e = data.find("\n", q0)
        if e < 0:
            e = n
        out.append((name, "".join(code.get(ch, "4") for ch in seq), data[q0:e]))
        p = e
    return out


@pytest.mark.parametrize("fq", ["wgsim100.fq", "ragged.fq", "short.fq", "sim_chr21_N100.fastq", "gapo.fq"])
def test_fastq_reader_on_cpu(built, golden, tmp_path, fq):
    """host/reads.c without a GPU: names (wgsim style, blanks), codes (lower case, N), qualities, '+name' lines, blank lines and
    a missing final newline, against a record-by-record restatement of the reference's reader."""
    out = tmp_path / "reads.tsv"
    run([bw.HOST_BIN, "dumpreads", os.path.join(golden, fq), str(out)])
    got = [tuple(ln.split("\t")) for ln in open(out).read().split("\n")[:-1]]
    assert got == _fastq2reads_model(os.path.join(golden, fq))
    # the Python-side loader used by the GPU tests sees the same codes on the plain 4-line files
    if fq != "wgsim100.fq":
        seqs, lens = bw.load_fastq_codes(os.path.join(golden, fq))
        assert ["".join(map(str, seqs[i, :lens[i]])) for i in range(len(lens))] == [g[1] for g in got]


@pytest.mark.parametrize("fq,chunk", [("ragged.fq", 7), ("wgsim100.fq", 1), ("short.fq", 1000), ("toy.fq", 64), ("sim_chr21_N100.fastq", 9)])
def test_streaming_fastq_reader_equals_the_whole_file_reader(built, golden, tmp_path, fq, chunk):
    """`align` takes the FASTQ in chunks (host/reads.c: fq_next_chunk, the same record scanner): whatever the chunk size, the reads and
    their codes are those of fastq2reads, and the padding beyond a read's length is code 4."""
    whole, parts = tmp_path / "whole.tsv", tmp_path / "parts.txt"
    run([bw.HOST_BIN, "dumpreads", os.path.join(golden, fq), str(whole)])
    run([bw.HOST_BIN, "dumpreads", os.path.join(golden, fq), str(parts), str(chunk)])
    assert [ln.split("\t")[1] for ln in open(whole).read().split("\n")[:-1]] == open(parts).read().split("\n")[:-1]


def _nasty_fastq(path, n=400, seed=9):
    """a FASTQ that a boundary guesser can misread: quality lines that start with '@' or '+', '@' and '+' inside names and qualities,
    '+name' separator lines, blank lines between records, lower-case bases, ragged lengths, no final newline"""
    import random
    rng = random.Random(seed)
    out = []
    for i in range(n):
        ln = rng.choice([1, 2, 17, 36, 100, 100, 151])
        seq = "".join(rng.choice("ACGTNacgt") for _ in range(ln))
        if rng.random() < 0.2:
            seq = "+" + seq[1:]  # (a base the reader encodes as N: after a quality line that starts with '@' the guess takes that line for a name)
        q0 = rng.choice("@+I@+5")
        qual = q0 + "".join(rng.choice("@+IJ5#") for _ in range(ln - 1))
        name = f"r{i}" + rng.choice(["", " @x", " +y", "/1 len=+@"])
        sep = "+" + (name if rng.random() < 0.3 else "")
        out.append(f"@{name}\n{seq}\n{sep}\n{qual}\n" + ("\n" if rng.random() < 0.1 else ""))
    text = "".join(out)
    open(path, "w").write(text[:-1] if text.endswith("\n") else text)


@pytest.mark.parametrize("region,threads,chunk", [("64", "3", 7), ("300", "7", 50), ("1000", "16", 1), ("4096", "5", 100000), ("1", "2", 33), ("100000000", "8", 13)])
def test_parallel_fastq_scanner_finds_the_sequential_scanners_records(built, golden, tmp_path, monkeypatch, region, threads, chunk):
    """host/reads.c (round 5): a region of the file is scanned by several threads that GUESS where their part's first record starts, and the
    parts are only accepted where the reference's sequential scan (io.c:430-498) would have arrived at the same '@'.  With regions of a
    few bytes to a few KB (BWB_FQ_REGION) every record is cut by part boundaries; on the golden files and on a FASTQ built to mislead the
    guess the chunks must hold exactly fastq2reads' reads."""
    nasty = tmp_path / "nasty.fq"
    _nasty_fastq(str(nasty))
    monkeypatch.setenv("BWB_FQ_REGION", region)
    monkeypatch.setenv("BWB_FQ_THREADS", threads)
    for fq in (os.path.join(golden, "wgsim100.fq"), os.path.join(golden, "ragged.fq"), os.path.join(golden, "sim_chr21_N100.fastq"), str(nasty)):
        whole, parts = tmp_path / "whole.tsv", tmp_path / "parts.txt"
        run([bw.HOST_BIN, "dumpreads", fq, str(whole)])
        run([bw.HOST_BIN, "dumpreads", fq, str(parts), str(chunk)])
        want = [ln.split("\t")[1] for ln in open(whole).read().split("\n")[:-1]]
        assert open(parts).read().split("\n")[:-1] == want, fq
        if fq == str(nasty):
            assert want == [g[1] for g in _fastq2reads_model(fq)] and len(want) == 400


def test_threaded_bwt_loader_reads_the_file_exactly(built, golden, tmp_path, monkeypatch):
    """host/bwt_io.c: the .bwt file read by several threads in units, blocks_ready monotone up to num_occ, the arrays (incl. the
    sampled SA) identical to the file's."""
    out = tmp_path / "copy.bwt"
    for threads in ("1", "5"):
        monkeypatch.setenv("BWB_LOAD_THREADS", threads)
        run([bw.HOST_BIN, "bwtcat", os.path.join(golden, "toy.fa.bwt"), str(out)])
        assert open(out, "rb").read() == open(os.path.join(golden, "toy.fa.bwt"), "rb").read()


@pytest.mark.parametrize("unit,threads", [("7", "5"), ("1", "16"), ("100", "3"), ("3129", "2"), ("3130", "4")])
def test_threaded_bwt_loader_with_many_small_units(built, golden, tmp_path, monkeypatch, unit, threads):
    """the multi-unit logic of host/bwt_io.c (units completing out of order, blocks_ready advancing over the leading complete units,
    the partial last unit) on the toy index: 3 130 blocks read in units of a few blocks (BWB_LOAD_UNIT) by up to 16 threads.  With the
    default unit (2^17 blocks) the toy index is one unit and only the GRCh37-size GPU tests ever ran this code."""
    out = tmp_path / "copy.bwt"
    monkeypatch.setenv("BWB_LOAD_THREADS", threads)
    monkeypatch.setenv("BWB_LOAD_UNIT", unit)
    run([bw.HOST_BIN, "bwtcat", os.path.join(golden, "toy.fa.bwt"), str(out)])
    assert open(out, "rb").read() == open(os.path.join(golden, "toy.fa.bwt"), "rb").read()


def test_bwt_loader_refuses_an_inconsistent_header(built, golden, tmp_path):
    """a header whose num_words does not belong to its length is refused before any loader thread sizes a read from it (a
    small num_words with a large num_occ made a unit's word count underflow and pread write past the array)"""
    import struct
    src = open(os.path.join(golden, "toy.fa.bwt"), "rb").read()
    hdr = list(struct.unpack("<5Q", src[:40]))
    for field, val in ((1, 16), (3, hdr[3] + 1000), (2, 1), (0, 1), (4, hdr[0])):
        bad = list(hdr)
        bad[field] = val
        path = tmp_path / f"bad{field}.bwt"
        open(path, "wb").write(struct.pack("<5Q", *bad) + src[40:])
        r = subprocess.run([bw.HOST_BIN, "bwtcat", str(path), str(tmp_path / "o.bwt")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode != 0 and "load_bwt" in r.stdout, (field, r.returncode, r.stdout[-200:])


def test_bwt_loader_refuses_a_header_that_promises_more_than_the_file_holds(built, golden, tmp_path):
    """a CONSISTENT header of a huge index in front of a small file is refused on the file's size, before any array is sized from it
    (the allocations used to come first - a crafted header drove an arbitrarily large malloc before it was rejected)"""
    import struct
    src = open(os.path.join(golden, "toy.fa.bwt"), "rb").read()
    length = 1 << 44  # 16 T characters: 2 TB of bwt words, 17 TB of O rows
    hdr = [length, (length + 7) // 8, (length + 31) // 32, (length + 127) // 128, 5]
    path = tmp_path / "huge.bwt"
    open(path, "wb").write(struct.pack("<5Q", *hdr) + src[40:])
    r = subprocess.run([bw.HOST_BIN, "bwtcat", str(path), str(tmp_path / "o.bwt")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "Could not read BWT" in r.stdout and "allocate" not in r.stdout, (r.returncode, r.stdout[-200:])


def test_aln_reader_refuses_a_record_longer_than_a_path(built, golden, tmp_path):
    """aln_length comes from the file and indexes a 272-byte path in the record builders: beyond the reference's 8-bit field it is refused"""
    import struct
    rec = struct.pack("<i", 1) + struct.pack("<iQQiiii", 9, 10, 12, 3, 0, 0, 60000) + struct.pack("<i", 1) + struct.pack("<i", 0 | (60000 << 2))
    path = tmp_path / "long.aln"
    open(path, "wb").write(rec)
    r = subprocess.run([bw.HOST_BIN, "alncat", str(path), str(tmp_path / "o.aln")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "not an ALN record" in r.stdout, (r.returncode, r.stdout[-200:])


@pytest.mark.parametrize("aln", ["toy_n0.aln", "toy_n3.aln", "toy_n4gap.aln", "ragged_n5.aln", "toy_s4gap.aln", "short_n2_t1.aln", "gapo_o6.aln"])
def test_aln_reader_and_writer_round_trip(built, golden, tmp_path, aln):
    """host/aln_io.c without a GPU.  The reference's loader fills aln_path in pair order (align.c:466-476), i.e. it holds the
    align-time path reversed - eval_aln and the CIGAR code work on that orientation (align.c:588-609) and so does ours.  So
    .aln -> alnsf2alns_bin -> alns2alnf_bin reverses the run-length pairs of a gapped record and leaves everything else alone,
    and doing it twice is the identity."""
    once, twice = tmp_path / "1.aln", tmp_path / "2.aln"
    run([bw.HOST_BIN, "alncat", os.path.join(golden, aln), str(once)])
    run([bw.HOST_BIN, "alncat", str(once), str(twice)])
    src = open(os.path.join(golden, aln), "rb").read()
    assert open(twice, "rb").read() == src
    import oracle_lib
    a, b = oracle_lib.parse_aln(src), oracle_lib.parse_aln(open(once, "rb").read())
    assert len(a) == len(b)
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb)
        for ea, eb in zip(ra, rb):
            assert {k: v for k, v in ea.items() if k != "states"} == {k: v for k, v in eb.items() if k != "states"}
            assert ea["states"] == eb["states"][::-1]


def test_interleave_helper_runs_the_command_or_says_why_not():
    """oracle/interleave_exec (bench.py's cpu_baseline: the reference under an interleaving NUMA policy on boxes without numactl):
    on a one-node machine it refuses with exit code 125 and runs nothing; with several nodes it execs the command."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "interleave_exec")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle"), exe], check=True)
    nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]) if os.path.isdir("/sys/devices/system/node") else 0
    r = subprocess.run([exe, "sh", "-c", "echo ran"], capture_output=True, text=True)
    if nodes >= 2 and r.returncode != 125:  # (125 also when the container forbids set_mempolicy)
        assert r.returncode == 0 and r.stdout.strip() == "ran"
    else:
        assert r.returncode == 125 and "ran" not in r.stdout
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.parametrize("aln", ["toy_n0.aln", "toy_n3.aln", "toy_n4gap.aln", "ragged_n5.aln", "toy_s4gap.aln", "short_n2_t1.aln", "gapo_o5.aln", "gapo_o6.aln", "himm_n5bigpen.aln", "sim_chr21_N100_n2.aln"])
@pytest.mark.parametrize("chunk", ["1", "7", "100000"])
def test_chunk_serialiser_writes_the_record_writers_bytes(built, golden, tmp_path, aln, chunk):
    """host/aln_io.c (round 5): `align` turns a chunk's hits into ONE byte buffer in the GPU worker (alns2alnf_buf, all cores) and the
    ordered writer only writes it; the bytes must be those of the record-by-record writer alns2alnf_bin (align.c:345-382) - all-match paths
    (the fast path), gapped paths with several runs, empty records, scores above 255."""
    one, buf = tmp_path / "one.aln", tmp_path / "buf.aln"
    run([bw.HOST_BIN, "alncat", os.path.join(golden, aln), str(one)])
    run([bw.HOST_BIN, "alncat", os.path.join(golden, aln), str(buf), "buf", chunk])
    assert open(buf, "rb").read() == open(one, "rb").read()

@pytest.mark.parametrize("team", ["1", "3", "64"])
def test_host_stages_do_not_depend_on_the_openmp_team(built, golden, tmp_path, monkeypatch, team):
    """host/bwb_host.h bwb_host_team(): the teams of the host stages are a clause on every region (at most 32 threads unless
    OMP_NUM_THREADS says otherwise - a team of 256 hardware threads was 4-30 x slower on the GPU box, profiles/r5_host_stage_threads.txt).
    Whatever the team, the reader's chunks and the serialiser's bytes are the same; `hostbench` reports both rates and the reader's stages."""
    import json
    monkeypatch.setenv("OMP_NUM_THREADS", team)
    fq, aln = os.path.join(golden, "sim_chr21_N100.fastq"), os.path.join(golden, "sim_chr21_N100_n2.aln")
    whole, parts, one, buf = tmp_path / "whole.tsv", tmp_path / "parts.txt", tmp_path / "one.aln", tmp_path / "buf.aln"
    run([bw.HOST_BIN, "dumpreads", fq, str(whole)])
    run([bw.HOST_BIN, "dumpreads", fq, str(parts), "17"])
    assert open(parts).read().split("\n")[:-1] == [ln.split("\t")[1] for ln in open(whole).read().split("\n")[:-1]]
    run([bw.HOST_BIN, "alncat", aln, str(one)])
    run([bw.HOST_BIN, "alncat", aln, str(buf), "buf", "13"])
    assert open(buf, "rb").read() == open(one, "rb").read()
    r = subprocess.run([bw.HOST_BIN, "hostbench", fq, aln], stdout=subprocess.PIPE, text=True, check=True)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["parse_reads"] == 100 and j["write_records"] == 100 and j["parse_reads_per_s"] > 0 and j["write_records_per_s"] > 0
    assert set(j["parse_stages_s"]) == {"scan", "alloc", "encode"}
