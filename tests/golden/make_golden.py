#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REAL reference (run in the build container only).

The reference (viq854/bwbble, /root/reference/mg-aligner) has no tests and no golden vectors
(SURVEY.md section 4), so parity is pinned by running the reference itself:

  * oracle/Makefile compiles the reference sources where they lie into oracle/_ref/bwbble;
  * this script drives that binary (`index`, `align -n ...`, `aln2sam`) on small synthetic inputs
    made by bwbble_amd/tools/bwb_synth.c, and
  * compiles a throw-away harness (written to a temp dir, never committed) that links the
    reference's own object files to dump known-answer vectors for O(), O_alphabet() and
    calculate_d().

Only DATA lands in tests/golden/: the inputs (FASTA/FASTQ), the reference-built index
(.bwt/.ann), the reference outputs (.aln/.sam) and the known-answer vectors (.npy).
No reference source text is copied.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/mg-aligner"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bwbble")
REF_OBJ = os.path.join(ROOT, "oracle", "_ref", "obj")

ALIGN_CONFIGS = {
    # name: extra flags for `bwbble align`
    "n0": ["-n", "0"],
    "n1": ["-n", "1"],
    "n2": ["-n", "2"],
    "n3": ["-n", "3"],
    "n5": ["-n", "5"],
    "n4gap": ["-n", "4", "-o", "2", "-e", "3", "-l", "20", "-k", "1"],
    "n2pen": ["-n", "2", "-M", "4", "-O", "6", "-E", "4", "-o", "2"],  # mm_score == gape_score bucket collision
    # single-genome mode (-S): 4-letter rank (O_actg_alphabet), 1-to-1 exact matching, children in A,G,C,T order
    "s0": ["-S", "-n", "0"],
    "s2": ["-S", "-n", "2"],
    "s4gap": ["-S", "-n", "4", "-o", "2", "-e", "3", "-l", "20", "-k", "1"],
    # precalculated 12-mer intervals (-P): heap seeded at readLen-12, reads with an N in the last 12 bases skipped.
    # (the reference builds toy.fa.pre, 74 MB, in ~4 minutes the first time; it is deleted again, only its SHA-256 is kept)
    "p0": ["-P", "-n", "0"],
    "p2": ["-P", "-n", "2"],
    "p4gap": ["-P", "-n", "4", "-o", "2", "-e", "3", "-l", "20", "-k", "1"],
}

HARNESS = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwt.h"
#include "align.h"
#include "inexact_match.h"
#include "io.h"
void calculate_d(bwt_t* BWT, char* read, const int readLen, diff_lower_bound_t* D, aln_params_t* params);
int main(int argc, char** argv) {
	bwt_t* BWT = load_bwt(argv[2], 0);
	if (strcmp(argv[1], "rank") == 0) {            /* rank <bwt> <pos.u64> <out.u64> */
		FILE* f = fopen(argv[3], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f) / 8; fseek(f, 0, SEEK_SET);
		bwtint_t* pos = malloc(8 * n); if (fread(pos, 8, n, f) != (size_t) n) return 2; fclose(f);
		FILE* o = fopen(argv[4], "wb");
		for (long q = 0; q < n; q++) {
			for (int inc = 0; inc < 2; inc++) {
				bwtint_t occ[16] = { 0 };
				O_alphabet(BWT, pos[q], 16, occ, inc);
				fwrite(occ, 8, 16, o);
			}
			bwtint_t single[16] = { 0 };
			for (int c = 1; c < 16; c++) single[c] = O(BWT, c, pos[q]);
			fwrite(single, 8, 16, o);
		}
		fclose(o);
	} else if (strcmp(argv[1], "dvec") == 0) {     /* dvec <bwt> <fastq> <out.i32> <seed_len> */
		reads_t* reads = fastq2reads(argv[3]);
		aln_params_t p; memset(&p, 0, sizeof p); set_default_aln_params(&p);
		int seed = atoi(argv[5]);
		FILE* o = fopen(argv[4], "wb");
		for (unsigned i = 0; i < reads->count; i++) {
			read_t* r = &reads->reads[i];
			diff_lower_bound_t* D = calloc(r->len + 1, sizeof *D);
			diff_lower_bound_t* Ds = calloc(seed + 1, sizeof *Ds);
			calculate_d(BWT, r->seq, r->len, D, &p);
			if (r->len > seed) calculate_d(BWT, r->seq, seed, Ds, &p);
			int hdr[2] = { r->len, seed };
			fwrite(hdr, 4, 2, o);
			fwrite(D, sizeof *D, r->len + 1, o);
			fwrite(Ds, sizeof *Ds, seed + 1, o);
			free(D); free(Ds);
		}
		fclose(o);
	}
	return 0;
}
"""


def run(cmd, **kw):
    print("+", " ".join(cmd))
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, **kw)


def wgsim_fixture(fq, out):
    """100 reads of toy.fq in the shape of the reference's own test_data/sim_chr21_N100.fastq (wgsim names with ':' and '/',
    bare '+' lines) plus what fastq2reads (io.c:410-515) also has to cope with: '+name' lines, lower-case bases, a name with
    blanks, blank lines between records and no newline at the end of the file."""
    lines = open(fq).read().split("\n")
    recs = [lines[i:i + 4] for i in range(0, 400, 4)]
    o = []
    for k, (name, seq, _, qual) in enumerate(recs):
        parts = name[1:].split("_")  # r<k>_chr<c>_<pos>_<strand>
        pos = int(parts[2])
        nm = f"{parts[1][3:]}_{pos}_{pos + 500 + k}_{k % 4}:0:0_{k % 3}:0:0_{k:x}/1"
        if k % 10 == 3:
            nm += " extra words"
        if k % 7 == 2:
            seq = seq.lower()
        plus = "+" + nm if k % 5 == 1 else "+"
        o.append(f"@{nm}\n{seq}\n{plus}\n{qual}")
        if k % 25 == 24:
            o.append("")
    open(out, "w").write("\n".join(o))


def extras():
    """Fixtures added in round 2 (kept separate so that they can be regenerated without the 4-minute -P table)."""
    fa, fq, fq2 = os.path.join(HERE, "toy.fa"), os.path.join(HERE, "toy.fq"), os.path.join(HERE, "ragged.fq")
    tmp = tempfile.mkdtemp(prefix="bwb_golden_")
    tfa = os.path.join(tmp, "toy.fa")
    for ext in ("", ".bwt", ".ann"):
        import shutil
        shutil.copy(fa + ext, tfa + ext)
    # config C5 of BASELINE.json (150 bp reads, -n 5 and the defaults -o 1 -e 6 -l 32 -k 2) on the ragged reads (36..150 bp)
    run([REF_BIN, "align"] + ALIGN_CONFIGS["n5"] + [tfa, fq2, os.path.join(HERE, "ragged_n5.aln")])
    # wgsim-shaped FASTQ through the reference's parser, aligner and SAM writer
    wq = os.path.join(HERE, "wgsim100.fq")
    wgsim_fixture(fq, wq)
    run([REF_BIN, "align", "-n", "2", tfa, wq, os.path.join(HERE, "wgsim100_n2.aln")])
    run([REF_BIN, "aln2sam", tfa, wq, os.path.join(HERE, "wgsim100_n2.aln"), os.path.join(HERE, "wgsim100_n2.sam")])
    # reads at or below the seed length mixed with longer ones, SERIAL reference (-t 1): a short read sees the D_seed bounds
    # the last longer read left in the buffer (inexact_match.c:33-35,62-65)
    synth = os.path.join(ROOT, "bwbble_amd", "bin", "bwb_synth")
    parts = []
    for k, (ln, cnt) in enumerate([(24, 6), (100, 5), (30, 8), (32, 8), (60, 4), (20, 10), (33, 6), (28, 10), (150, 3), (31, 12)]):
        q = os.path.join(tmp, f"s{k}.fq")
        run([synth, "reads", fa, q, str(cnt), str(ln), str(300 + k), "3.0", "5.0", "5.0"])
        parts.append(open(q).read())
    sq = os.path.join(HERE, "short.fq")
    open(sq, "w").write("".join(parts))
    for name, flags in (("n2", ["-n", "2"]), ("n3k1", ["-n", "3", "-k", "1"]), ("p2", ["-P", "-n", "2"])):
        if "-P" in flags and not os.path.exists(tfa + ".pre"):
            print("(the reference builds the 74 MB .pre table first: about 4 minutes)")
        run([REF_BIN, "align"] + flags + ["-t", "1", tfa, sq, os.path.join(HERE, f"short_{name}_t1.aln")])


def round4():
    """Round 4: parameters whose entry scores pass 255 (aln_entry_t.score is an 8-bit field, align.h:104: the break test at
    inexact_match.c:309 sees the score modulo 256, while aln_t.score - recomputed at :332,348 - is the full int): mismatch-rich reads
    (4 % substitutions), -n 5 -M 52 -O 60 -E 30 -> 642 heap buckets, five mismatches score 260."""
    fa = os.path.join(HERE, "toy.fa")
    tmp = tempfile.mkdtemp(prefix="bwb_golden_")
    tfa = os.path.join(tmp, "toy.fa")
    import shutil
    for ext in ("", ".bwt", ".ann"):
        shutil.copy(fa + ext, tfa + ext)
    synth = os.path.join(ROOT, "bwbble_amd", "bin", "bwb_synth")
    hq = os.path.join(HERE, "himm.fq")
    run([synth, "reads", fa, hq, "160", "100", "77", "4.0", "10.0", "2.0"])
    run([REF_BIN, "align", "-n", "5", "-M", "52", "-O", "60", "-E", "30", tfa, hq, os.path.join(HERE, "himm_n5bigpen.aln")])


def round5():
    """Round 5.  (1) The reference's OWN test input, test_data/sim_chr21_N100.fastq (data: 100 wgsim reads of chr21; config C1 of
    BASELINE.json names it), through the reference's parser, aligner and SAM writer on the toy index - real chr21 reads do not map to a
    synthetic text, so the records are nearly all empty: a parser / plumbing fixture.  (2) Parameters beyond what round 4's library
    accepted: more than four gap opens per alignment (-o 5: aln_entry_t's 256-byte path holds any number, align.h:100-119) and
    penalties above 63 (heap buckets are exact scores, inexact_match.c:510-516,548-591)."""
    import shutil
    fa = os.path.join(HERE, "toy.fa")
    tmp = tempfile.mkdtemp(prefix="bwb_golden_")
    tfa = os.path.join(tmp, "toy.fa")
    for ext in ("", ".bwt", ".ann"):
        shutil.copy(fa + ext, tfa + ext)
    src = "/root/reference/test_data/sim_chr21_N100.fastq"
    cq = os.path.join(HERE, "sim_chr21_N100.fastq")
    shutil.copy(src, cq)
    os.chmod(cq, 0o644)
    for name, flags in (("n0", ["-n", "0"]), ("n2", ["-n", "2"])):
        run([REF_BIN, "align"] + flags + [tfa, cq, os.path.join(HERE, f"sim_chr21_N100_{name}.aln")])
    run([REF_BIN, "aln2sam", tfa, cq, os.path.join(HERE, "sim_chr21_N100_n2.aln"), os.path.join(HERE, "sim_chr21_N100_n2.sam")])
    hq = os.path.join(HERE, "himm.fq")  # (round 4's mismatch- and indel-rich reads)
    # reads with 2..6 single-base deletions or insertions, 12+ bases apart, cut from A/C/G/T-only windows of the toy genome: their best
    # alignments open that many gaps (bwb_synth puts at most one indel into a read)
    import random
    rng = random.Random(505)
    seq = "".join(ln.strip() for ln in open(fa).read().split(">")[1].split("\n")[1:])
    recs = []
    while len(recs) < 36:
        k = 2 + len(recs) % 5
        start = rng.randrange(0, len(seq) - 130)
        win = seq[start:start + 120]
        if set(win) - set("ACGT"):
            continue
        pos = sorted(rng.sample(range(18, 96, 13), k))
        out, dele = [], len(recs) % 2 == 0
        for i, ch in enumerate(win):
            if i in pos:
                if dele:
                    continue
                out.append(ch + rng.choice("ACGT"))
            else:
                out.append(ch)
        read = "".join(out)[:100]
        if len(recs) % 3 == 2:  # reverse strand
            read = read[::-1].translate(str.maketrans("ACGT", "TGCA"))
        recs.append(f"@gap{len(recs)}_{k}{'del' if dele else 'ins'}\n{read}\n+\n{'2' * len(read)}\n")
    gq = os.path.join(HERE, "gapo.fq")
    open(gq, "w").write("".join(recs))
    run([REF_BIN, "align", "-n", "6", "-o", "6", "-e", "6", "-m", "200000", tfa, gq, os.path.join(HERE, "gapo_o6.aln")])
    run([REF_BIN, "align", "-n", "5", "-o", "5", "-e", "2", tfa, gq, os.path.join(HERE, "gapo_o5.aln")])
    run([REF_BIN, "align", "-n", "3", "-M", "80", "-O", "90", "-E", "70", tfa, hq, os.path.join(HERE, "himm_M80.aln")])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--extras":
        return extras()
    if len(sys.argv) > 1 and sys.argv[1] == "--round4":
        return round4()
    if not os.path.isdir(REF_SRC):
        sys.exit("reference sources not present; golden vectors can only be regenerated in the build container")
    run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp(prefix="bwb_golden_")
    synth = os.path.join(tmp, "bwb_synth")
    run(["gcc", "-O2", "-o", synth, os.path.join(ROOT, "bwbble_amd", "tools", "bwb_synth.c")])

    fa, fq = os.path.join(HERE, "toy.fa"), os.path.join(HERE, "toy.fq")
    # toy multi-genome: 200 000 fwd chars in 3 records + 4 indel bubbles
    run([synth, "genome", fa, "200000", "3", "4", "7"])
    # 600 x 100 bp reads, 1 % substitutions, 2 % of reads with an indel, 1 % of reads with an N
    run([synth, "reads", fa, fq, "600", "100", "11", "1.0", "2.0", "1.0"])
    # ragged lengths (36..150, crosses the READ_LENGTH_ALLOC=150 and seed_length=32 edges) incl. N-rich reads
    fq2 = os.path.join(HERE, "ragged.fq")
    parts = []
    for k, (ln, cnt) in enumerate([(36, 40), (51, 40), (75, 40), (125, 40), (150, 40)]):
        p = os.path.join(tmp, f"r{k}.fq")
        run([synth, "reads", fa, p, str(cnt), str(ln), str(100 + k), "2.0", "5.0", "20.0"])
        parts.append(open(p).read())
    open(fq2, "w").write("".join(parts))

    run([REF_BIN, "index", fa])
    os.remove(fa + ".ref")  # 400 kB of raw text, not needed by align/aln2sam
    for name, flags in ALIGN_CONFIGS.items():
        run([REF_BIN, "align"] + flags + [fa, fq, os.path.join(HERE, f"toy_{name}.aln")])
    for name in ("n0", "n3", "n4gap", "s2", "p2"):
        run([REF_BIN, "align"] + ALIGN_CONFIGS[name] + [fa, fq2, os.path.join(HERE, f"ragged_{name}.aln")])
    if os.path.exists(fa + ".pre"):
        import hashlib
        hsh = hashlib.sha256()
        with open(fa + ".pre", "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                hsh.update(blk)
        open(os.path.join(HERE, "toy.fa.pre.sha256"), "w").write(hsh.hexdigest() + "\n")
        os.remove(fa + ".pre")
    run([REF_BIN, "aln2sam", fa, fq, os.path.join(HERE, "toy_n3.aln"), os.path.join(HERE, "toy_n3.sam")])
    run([REF_BIN, "aln2sam", fa, fq2, os.path.join(HERE, "ragged_n4gap.aln"), os.path.join(HERE, "ragged_n4gap.sam")])

    # known-answer vectors through a harness linked against the reference's own objects
    hsrc = os.path.join(tmp, "harness.c")
    open(hsrc, "w").write(HARNESS)
    objs = [os.path.join(REF_OBJ, o) for o in os.listdir(REF_OBJ) if o.endswith(".o") and o != "main.o"]
    hbin = os.path.join(tmp, "harness")
    run(["gcc", "-w", "-O2", "-std=gnu99", "-fopenmp", "-I", REF_SRC, hsrc] + objs + ["-o", hbin, "-lm"])

    hdr = np.fromfile(fa + ".bwt", dtype=np.uint64, count=5)
    length, num_words, sa0 = int(hdr[0]), int(hdr[1]), int(hdr[4])
    words = np.fromfile(fa + ".bwt", dtype=np.uint32, count=num_words, offset=8 * 22)
    first = (words[::16] >> 28).astype(np.int64)  # first char of every 128-char block
    rng = np.random.default_rng(5)
    pos = [2**64 - 1, length - 1, 0, 1, 127, 128, 129, length - 2, sa0, sa0 - 1, sa0 + 1, (sa0 // 128) * 128]
    for code in (5, 9, 11, 13, 0, 10):  # blocks starting with an uncounted 3-base code, '$' and N
        blocks = np.nonzero(first == code)[0][:6]
        for b in blocks:
            pos += [int(b) * 128, int(b) * 128 + 1, min(int(b) * 128 + 127, length - 1), min(int(b) * 128 + 60, length - 1)]
    nblk = (length + 127) // 128
    for b in rng.integers(0, nblk, 300):
        pos += [int(b) * 128, max(int(b) * 128 - 1, 0)]
    pos += [int(v) for v in rng.integers(0, length, 6000)]
    pos = np.array([p for p in pos if p == 2**64 - 1 or 0 <= p < length], dtype=np.uint64)
    ppath, opath = os.path.join(tmp, "pos.u64"), os.path.join(tmp, "rank.u64")
    pos.tofile(ppath)
    run([hbin, "rank", fa + ".bwt", ppath, opath])
    out = np.fromfile(opath, dtype=np.uint64).reshape(len(pos), 3, 16)
    np.save(os.path.join(HERE, "rank_pos.npy"), pos)
    np.save(os.path.join(HERE, "rank_O_alphabet_inc0.npy"), out[:, 0, :])
    np.save(os.path.join(HERE, "rank_O_alphabet_inc1.npy"), out[:, 1, :])
    np.save(os.path.join(HERE, "rank_O_single.npy"), out[:, 2, :])

    for tag, path in (("toy", fq), ("ragged", fq2)):
        dpath = os.path.join(tmp, f"d_{tag}.i32")
        run([hbin, "dvec", fa + ".bwt", path, dpath, "32"])
        np.save(os.path.join(HERE, f"dvec_{tag}.npy"), np.fromfile(dpath, dtype=np.int32))
    extras()
    print("golden vectors written to", HERE)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "round5":
    round5()
    sys.exit(0)
if __name__ == "__main__":
    main()
