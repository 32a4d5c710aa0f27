/*
 * bwb_synth - deterministic synthetic multi-genome + read generator (test/bench tooling).
 *
 * There is no real chr21 / GRCh37 / 1000G data offline (SURVEY.md header), so every workload is
 * synthetic.  The FASTA has the shape that the reference's mg-ref/comb tool emits
 * (mg-ref/comb.cpp:121-145 IUPAC codes at SNP sites, :244-267 ">bubbleK chr pos" records with
 * 124-char flanks), and the FASTQ has the 4-line shape that fastq2reads parses
 * (mg-aligner/io.c:410-515).
 *
 *   bwb_synth genome <out.fasta> <n_fwd_chars> <n_records> <n_bubbles> <seed>
 *   bwb_synth reads  <in.fasta> <out.fastq> <n_reads> <read_len> <seed> [sub_rate_pct=1.0] [indel_read_pct=0.1] [n_read_pct=0.0]
 *
 * Everything is derived from splitmix64/xoshiro256** streams seeded by <seed>, so the same
 * arguments give byte-identical files on any machine.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s[4]; } rng_t;
static uint64_t splitmix64(uint64_t *x) {
	uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r) {
	uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
	s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
	return res;
}
static inline uint64_t rng_below(rng_t *r, uint64_t n) { return (uint64_t)(((__uint128_t)rng_next(r) * n) >> 64); }
static inline double rng_unit(rng_t *r) { return (rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* base-set bitmask (A=1,C=2,G=4,T=8) -> IUPAC letter */
static const char iupac_of_mask[16] = { 'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N' };
static int mask_of_iupac(char c) {
	for (int m = 1; m < 16; m++) if (iupac_of_mask[m] == c) return m;
	return 15;
}
static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
static int idx_of_base(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
static char compl_base(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

static void die(const char *m) { fprintf(stderr, "bwb_synth: %s\n", m); exit(1); }

static void write_wrapped(FILE *f, const char *s, uint64_t n) {
	for (uint64_t i = 0; i < n; i += 60) {
		uint64_t k = n - i < 60 ? n - i : 60;
		fwrite(s + i, 1, k, f);
		fputc('\n', f);
	}
}

#define N_FAM 50
#define FAM_LEN 300
#define FLANK 124

static int cmd_genome(int argc, char **argv) {
	if (argc < 7) die("usage: genome <out.fasta> <n_fwd_chars> <n_records> <n_bubbles> <seed>");
	const char *out = argv[2];
	uint64_t n = strtoull(argv[3], 0, 10);
	int n_rec = atoi(argv[4]);
	uint64_t n_bub = strtoull(argv[5], 0, 10);
	uint64_t seed = strtoull(argv[6], 0, 10);
	if (n_rec < 1 || n < (uint64_t)n_rec * 1000) die("genome too small for the record count");
	rng_t r; rng_seed(&r, seed);

	/* hidden haploid sequence: 90 % iid, 10 % copies of 50 repeat families at 5-15 % divergence */
	char *hid = (char *)malloc(n), *txt = (char *)malloc(n);
	if (!hid || !txt) die("out of memory");
	char fam[N_FAM][FAM_LEN];
	for (int f = 0; f < N_FAM; f++) for (int i = 0; i < FAM_LEN; i++) fam[f][i] = ACGT[rng_below(&r, 4)];
	uint64_t i = 0;
	while (i < n) {
		if (rng_below(&r, 3000) < 1 && i + FAM_LEN <= n) { /* 300/3000 = 10 % of the positions */
			int f = (int)rng_below(&r, N_FAM);
			double div = 0.05 + 0.10 * rng_unit(&r);
			for (int k = 0; k < FAM_LEN; k++) {
				char c = fam[f][k];
				if (rng_unit(&r) < div) c = ACGT[(idx_of_base(c) + 1 + rng_below(&r, 3)) & 3];
				hid[i++] = c;
			}
		} else hid[i++] = ACGT[rng_below(&r, 4)];
	}
	/* IUPAC text: 1.2 % two-base codes (hidden base + one other), 0.01 % three-base, 0.005 % N */
	for (i = 0; i < n; i++) {
		int m = 1 << idx_of_base(hid[i]);
		uint64_t u = rng_below(&r, 1000000);
		if (u < 12000) m |= 1 << ((idx_of_base(hid[i]) + 1 + rng_below(&r, 3)) & 3);
		else if (u < 12100) m = 15 ^ (1 << ((idx_of_base(hid[i]) + 1 + rng_below(&r, 3)) & 3));
		else if (u < 12150) m = 15;
		txt[i] = iupac_of_mask[m];
	}
	/* records sized proportionally to 1/(k+1)^0.35 (roughly the GRCh37 chromosome size spread) */
	uint64_t *rstart = (uint64_t *)malloc(sizeof(uint64_t) * (n_rec + 1));
	double tot = 0, acc = 0;
	for (int k = 0; k < n_rec; k++) { double w = 1.0; for (int q = 0; q < k; q++) w *= 0.965; tot += w; }
	rstart[0] = 0;
	for (int k = 0; k < n_rec; k++) {
		double w = 1.0; for (int q = 0; q < k; q++) w *= 0.965;
		acc += w;
		rstart[k + 1] = (k == n_rec - 1) ? n : (uint64_t)((double)n * acc / tot);
	}
	FILE *f = fopen(out, "w");
	if (!f) die("cannot open output FASTA");
	for (int k = 0; k < n_rec; k++) {
		fprintf(f, ">chr%d synthetic seed=%llu\n", k + 1, (unsigned long long)seed);
		write_wrapped(f, txt + rstart[k], rstart[k + 1] - rstart[k]);
	}
	/* indel bubbles: 124 flank + 1-5 inserted bases + 124 flank, header ">bubbleK chr pos" */
	char buf[2 * FLANK + 8];
	for (uint64_t b = 0; b < n_bub; b++) {
		int k = (int)rng_below(&r, n_rec);
		uint64_t len = rstart[k + 1] - rstart[k];
		if (len < 4 * FLANK) { k = 0; len = rstart[1] - rstart[0]; }
		uint64_t pos = FLANK + rng_below(&r, len - 2 * FLANK);
		int ins = 1 + (int)rng_below(&r, 5);
		memcpy(buf, txt + rstart[k] + pos - FLANK, FLANK);
		for (int q = 0; q < ins; q++) buf[FLANK + q] = ACGT[rng_below(&r, 4)];
		memcpy(buf + FLANK + ins, txt + rstart[k] + pos, FLANK);
		fprintf(f, ">bubble%llu chr%d %llu\n", (unsigned long long)b, k + 1, (unsigned long long)pos);
		write_wrapped(f, buf, 2 * FLANK + ins);
	}
	fclose(f);
	free(hid); free(txt); free(rstart);
	return 0;
}

/* load the first-class ">chr" records of a FASTA (bubbles are skipped as read sources) */
static char *load_chr_text(const char *fa, uint64_t *n_out, uint64_t **starts_out, int *nrec_out) {
	FILE *f = fopen(fa, "r");
	if (!f) die("cannot open input FASTA");
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	char *raw = (char *)malloc(sz + 1);
	if (fread(raw, 1, sz, f) != (size_t)sz) die("short read on FASTA");
	fclose(f);
	char *txt = (char *)malloc(sz + 1);
	uint64_t n = 0; int nrec = 0, cap = 64, take = 0;
	uint64_t *starts = (uint64_t *)malloc(sizeof(uint64_t) * cap);
	for (long p = 0; p < sz;) {
		if (raw[p] == '>') {
			take = strncmp(raw + p, ">chr", 4) == 0;
			if (take) { if (nrec + 2 > cap) { cap *= 2; starts = (uint64_t *)realloc(starts, sizeof(uint64_t) * cap); } starts[nrec++] = n; }
			while (p < sz && raw[p] != '\n') p++;
			p++;
		} else { if (take && raw[p] != '\n') txt[n++] = raw[p]; p++; }
	}
	starts[nrec] = n;
	free(raw);
	*n_out = n; *starts_out = starts; *nrec_out = nrec;
	return txt;
}

static int cmd_reads(int argc, char **argv) {
	if (argc < 7) die("usage: reads <in.fasta> <out.fastq> <n_reads> <read_len> <seed> [sub_pct] [indel_read_pct] [n_read_pct]");
	uint64_t n; uint64_t *starts; int nrec;
	char *txt = load_chr_text(argv[2], &n, &starts, &nrec);
	uint64_t n_reads = strtoull(argv[4], 0, 10);
	int len = atoi(argv[5]);
	uint64_t seed = strtoull(argv[6], 0, 10);
	double sub = (argc > 7 ? atof(argv[7]) : 1.0) / 100.0;
	double indel = (argc > 8 ? atof(argv[8]) : 0.1) / 100.0;
	double nread = (argc > 9 ? atof(argv[9]) : 0.0) / 100.0;
	if (len < 8 || len > 250) die("read_len must be in [8,250]");
	rng_t r; rng_seed(&r, seed ^ 0xC0FFEEull);
	FILE *f = fopen(argv[3], "w");
	if (!f) die("cannot open output FASTQ");
	char *buf = (char *)malloc(len + 8), *rc = (char *)malloc(len + 8), *q = (char *)malloc(len + 8);
	memset(q, '2', len); q[len] = 0;
	for (uint64_t k = 0; k < n_reads; k++) {
		int rec; uint64_t rl;
		do { rec = (int)rng_below(&r, nrec); rl = starts[rec + 1] - starts[rec]; } while (rl < (uint64_t)len + 8);
		uint64_t pos = rng_below(&r, rl - len - 4);
		const char *src = txt + starts[rec] + pos;
		/* resolve every IUPAC code to one of its bases (so SNP alleles are exercised) */
		int has_indel = rng_unit(&r) < indel, ipos = 5 + (int)rng_below(&r, len - 10), ikind = (int)rng_below(&r, 2);
		int o = 0;
		for (int s = 0; o < len; s++) {
			int m = mask_of_iupac(src[s]);
			int cand[4], nc = 0;
			for (int b = 0; b < 4; b++) if (m & (1 << b)) cand[nc++] = b;
			char c = ACGT[cand[rng_below(&r, nc)]];
			if (has_indel && o == ipos) {
				has_indel = 0;
				if (ikind == 0) { buf[o++] = ACGT[rng_below(&r, 4)]; s--; continue; } /* insertion in the read */
				else continue;                                                        /* deletion from the read */
			}
			if (rng_unit(&r) < sub) c = ACGT[(idx_of_base(c) + 1 + rng_below(&r, 3)) & 3];
			buf[o++] = c;
		}
		if (rng_unit(&r) < nread) buf[rng_below(&r, len)] = 'N';
		buf[len] = 0;
		int strand = (int)rng_below(&r, 2);
		if (strand) { for (int j = 0; j < len; j++) rc[j] = compl_base(buf[len - 1 - j]); rc[len] = 0; }
		fprintf(f, "@r%llu_chr%d_%llu_%c\n%s\n+\n%s\n", (unsigned long long)k, rec + 1, (unsigned long long)(pos + 1), strand ? '-' : '+', strand ? rc : buf, q);
	}
	fclose(f);
	free(buf); free(rc); free(q); free(txt); free(starts);
	return 0;
}

int main(int argc, char **argv) {
	if (argc >= 2 && strcmp(argv[1], "genome") == 0) return cmd_genome(argc, argv);
	if (argc >= 2 && strcmp(argv[1], "reads") == 0) return cmd_reads(argc, argv);
	die("commands: genome | reads");
	return 1;
}
