/*
 * index.c - `bwbble index`: FASTA -> .ref/.ann/.bwt, byte-compatible with the reference
 * (mg-aligner/bwt.c:29-63,161-218,266-291; io.c:190-321).
 *
 * Host plumbing, not the accelerated path.  The reference builds the suffix array with a vendored
 * sais-lite (is.c); this file uses its own construction: suffixes are distributed by their first
 * five characters (20-bit radix, counting sort) and every bucket is refined by sorting 16-character
 * super-characters (64-bit keys) recursively, buckets in parallel (OpenMP).  The resulting order is
 * the plain suffix order with the end of the text smaller than every character, which is what
 * is_bwt() produces (is.c:197-243), so the files come out identical (tests/test_host_tools.py).  Long exact repeats (N runs
 * of real assemblies, homopolymers, satellites), on which 16 characters per pass would be quadratic, are handed to a prefix
 * doubling phase (finish_deep_groups); the scratch of a pass is sized for the group at hand, not for the largest bucket.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bwb_host.h"

/* io.h:32 and io.h:132-149 */
static const unsigned char iupacCompl[16] = { 0, 15, 8, 7, 4, 11, 12, 3, 2, 13, 10, 5, 6, 9, 14, 1 };
static unsigned char nt16(int c) {
	switch (c) {
	case '$': return 0;
	case 'T': return 1; case 'K': return 2; case 'G': return 3; case 'S': return 4; case 'B': return 5; case 'Y': return 6;
	case 'C': return 7; case 'M': return 8; case 'H': return 9; case 'N': return 10; case 'V': return 11; case 'R': return 12;
	case 'D': return 13; case 'W': return 14; case 'A': return 15;
	default: return 10; /* anything else is N */
	}
}

static double wall(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* io.c:190-321 */
void fasta2ref(const char *fastaFname, const char *refFname, const char *annFname, unsigned char **seq_out, bwtint_t *totalSeqLen) {
	FILE *f = fopen(fastaFname, "r");
	if (!f) bwb_die("fasta2ref: Cannot open FASTA file: %s!", fastaFname);
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	char *raw = (char *)malloc((size_t)sz + 1);
	if (!raw || fread(raw, 1, (size_t)sz, f) != (size_t)sz) bwb_die("fasta2ref: Cannot read FASTA file: %s!", fastaFname);
	fclose(f);
	if (sz == 0 || raw[0] != '>') bwb_die("Error: File %s not in FASTA format", fastaFname);
	FILE *annFile = fopen(annFname, "wb");
	if (!annFile) bwb_die("fasta2ref: Cannot open .ann file: %s!", annFname);

	unsigned char *seq = (unsigned char *)malloc(2 * (size_t)sz + 16);
	if (!seq) bwb_die("fasta2ref: Could not allocate memory for the input sequence");
	bwtint_t seqLen = 0;
	int nann = 0, cap = 256;
	seq_annotation_t *anns = (seq_annotation_t *)malloc(sizeof(seq_annotation_t) * (size_t)cap);
	long p = 1; /* past the first '>' */
	while (p <= sz) {
		if (nann == cap) { cap *= 2; anns = (seq_annotation_t *)realloc(anns, sizeof(seq_annotation_t) * (size_t)cap); }
		seq_annotation_t *a = &anns[nann];
		int nl = 0;
		while (p < sz && raw[p] != '\n' && nl < MAX_SEQ_NAME_LEN) a->name[nl++] = raw[p++]; /* io.c:237-241 */
		a->name[nl] = 0;
		while (p < sz && raw[p] != '\n') p++;
		if (p >= sz) bwb_die("Error: File %s not in FASTA format", fastaFname);      /* io.c:246 */
		bwtint_t sub = 0;
		p++;
		while (p < sz && raw[p] != '>') {                                             /* io.c:250-273 */
			int c = raw[p++];
			if (c == '\n') continue;
			if (c >= 'a' && c <= 'z') c += 'A' - 'a';
			seq[seqLen++] = nt16(c);
			sub++;
		}
		seq[seqLen++] = 0; /* '$' separator after every record, io.c:276 */
		sub++;
		printf("Done reading a subsequence of size %llu from FASTA\n", (unsigned long long)sub);
		a->start_index = seqLen - sub;
		a->end_index = seqLen - 1;
		nann++;
		if (p >= sz) break;
		p++; /* skip '>' */
	}
	printf("Done reading FASTA file. Total sequence length read = %llu\n", (unsigned long long)seqLen);
	fprintf(annFile, "%llu\t%d\n", (unsigned long long)seqLen, nann);                 /* io.c:292-296 */
	for (int i = 0; i < nann; i++)
		fprintf(annFile, "%s\t%llu\t%llu\n", anns[i].name, (unsigned long long)anns[i].start_index, (unsigned long long)anns[i].end_index);
	fclose(annFile);
	for (bwtint_t i = 0; i < seqLen; i++) seq[2 * seqLen - i - 1] = iupacCompl[seq[i]]; /* io.c:304-306 */
	*totalSeqLen = 2 * seqLen;
	if (refFname) {
		FILE *rf = fopen(refFname, "wb");
		if (!rf) bwb_die("fasta2ref: Cannot open .ref file: %s!", refFname);
		fwrite(seq, 1, 2 * seqLen, rf);                                               /* io.c:269,311: raw codes, fwd then revcomp */
		fclose(rf);
		printf("Wrote %llu chars to ref file\n", (unsigned long long)(2 * seqLen));
	}
	free(anns);
	free(raw);
	*seq_out = seq;
}

fasta_annotations_t *annf2ann(const char *annFname) { /* io.c:324-349 */
	FILE *f = fopen(annFname, "r");
	if (!f) bwb_die("annf2ann: Cannot open ANN file: %s!", annFname);
	fasta_annotations_t *a = (fasta_annotations_t *)calloc(1, sizeof(*a));
	unsigned long long tot;
	if (fscanf(f, "%llu\t%d\n", &tot, &a->num_seq) != 2) bwb_die("annf2ann: Could not parse ANN file: %s!", annFname);
	a->seq_anns = (seq_annotation_t *)calloc((size_t)a->num_seq, sizeof(seq_annotation_t));
	for (int i = 0; i < a->num_seq; i++) {
		unsigned long long s, e;
		if (fscanf(f, "%256[^\n\t]\t%llu\t%llu\n", a->seq_anns[i].name, &s, &e) < 3) bwb_die("annf2ann: Could not parse ANN file: %s!", annFname);
		a->seq_anns[i].start_index = s; a->seq_anns[i].end_index = e;
	}
	fclose(f);
	return a;
}
void free_ann(fasta_annotations_t *a) { if (a) { free(a->seq_anns); free(a); } }

/* ---------------------------------------------------------------------------------------------
 * Suffix sorting
 * ------------------------------------------------------------------------------------------- */
typedef struct { uint64_t key; uint64_t idx; } kv_t;

/* 16 characters starting at position i as one big-endian 64-bit key; pk = nibble-packed text
 * (char 2k in the high nibble of byte k) padded with zero bytes. */
static inline uint64_t key16(const unsigned char *pk, uint64_t i) {
	uint64_t w;
	memcpy(&w, pk + (i >> 1), 8);
	w = __builtin_bswap64(w);
	if (i & 1) w = (w << 4) | (pk[(i >> 1) + 8] >> 4);
	return w;
}

static void kv_insertion(kv_t *a, long n) {
	for (long i = 1; i < n; i++) {
		kv_t v = a[i];
		long j = i - 1;
		while (j >= 0 && a[j].key > v.key) { a[j + 1] = a[j]; j--; }
		a[j + 1] = v;
	}
}
static void kv_sort(kv_t *a, long n) { /* quicksort on key, median of three, explicit stack */
	long stk[128][2];
	int sp = 0;
	stk[sp][0] = 0; stk[sp][1] = n - 1; sp++;
	while (sp) {
		sp--;
		long lo = stk[sp][0], hi = stk[sp][1];
		while (hi - lo > 24) {
			long mid = lo + (hi - lo) / 2;
			uint64_t x = a[lo].key, y = a[mid].key, z = a[hi].key;
			uint64_t pv = (x < y) ? ((y < z) ? y : (x < z ? z : x)) : ((x < z) ? x : (y < z ? z : y));
			long i = lo, j = hi;
			while (i <= j) {
				while (a[i].key < pv) i++;
				while (a[j].key > pv) j--;
				if (i <= j) { kv_t t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
			}
			if (j - lo < hi - i) { if (i < hi) { stk[sp][0] = i; stk[sp][1] = hi; sp++; } hi = j; }
			else { if (lo < j) { stk[sp][0] = lo; stk[sp][1] = j; sp++; } lo = i; }
		}
		kv_insertion(a + lo, hi - lo + 1);
	}
}

/* Groups the key refinement gives up on - a long exact repeat (an N run, a homopolymer, a satellite array, a duplicated
 * segment), where advancing 16 characters per pass would cost (repeat length)^2 / 32 - are finished by prefix doubling. */
#define DEPTH_MAX 512            /* characters after which a group still unsorted is handed over */
#define STALL_MIN 4096           /* ... or earlier, when a group this large keeps (almost) all its members through a pass */
typedef struct { uint64_t start, m, depth; } deep_t;
static deep_t *g_deep; static size_t g_ndeep, g_capdeep;
static void deep_add(uint64_t start, uint64_t m, uint64_t depth) {
#pragma omp critical(bwb_deep)
	{
		if (g_ndeep == g_capdeep) {
			g_capdeep = g_capdeep ? 2 * g_capdeep : 1024;
			g_deep = (deep_t *)realloc(g_deep, g_capdeep * sizeof(deep_t));
			if (!g_deep) bwb_die("is_bwt: out of memory");
		}
		g_deep[g_ndeep++] = (deep_t){ start, m, depth };
	}
}
static kv_t *kv_alloc(long m) {
	kv_t *t = (kv_t *)malloc(sizeof(kv_t) * (size_t)(m > 0 ? m : 1));
	if (!t) bwb_die("is_bwt: Could not allocate memory for the BWT index construction, alloc'ing %llu Mb", (unsigned long long)((size_t)m * sizeof(kv_t) >> 20));
	return t;
}

/* sorts the suffixes sa[0..m) that agree on their first `depth` characters; sa0 = start of the whole array (for deep_add) */
static void refine(const unsigned char *pk, uint64_t n, uint64_t *sa0, uint64_t *sa, long m, uint64_t depth) {
	kv_t small[48], *tmp = small, *heap = NULL; /* most groups are tiny: no allocation for them */
	long tmp_cap = 48;
	int stalled = 0;
	while (m > 1) {
		/* suffixes that end exactly here are the smallest of the group; among them the shorter (= larger
		 * start) comes first */
		long ne = 0;
		for (long i = 0; i < m; i++)
			if (sa[i] + depth >= n) { uint64_t t = sa[i]; sa[i] = sa[ne]; sa[ne] = t; ne++; }
		for (long i = 1; i < ne; i++) { uint64_t v = sa[i]; long j = i - 1; while (j >= 0 && sa[j] < v) { sa[j + 1] = sa[j]; j--; } sa[j + 1] = v; }
		sa += ne; m -= ne;
		if (m <= 1) break;
		/* (every member left has more than `depth` characters, so the doubling phase never looks past the end of the text) */
		if (stalled || depth >= DEPTH_MAX) { deep_add((uint64_t)(sa - sa0), (uint64_t)m, depth); break; }
		if (m > tmp_cap) { free(heap); tmp = heap = kv_alloc(m); tmp_cap = m; } /* sized for this group (groups only shrink), not for the largest bucket */
		for (long i = 0; i < m; i++) { tmp[i].key = key16(pk, sa[i] + depth); tmp[i].idx = sa[i]; }
		kv_sort(tmp, m);
		for (long i = 0; i < m; i++) sa[i] = tmp[i].idx;
		/* recurse into groups of equal keys; the largest is handled by the loop */
		long g0 = 0, big0 = -1, bigm = 0;
		for (long i = 1; i <= m; i++) {
			if (i == m || tmp[i].key != tmp[g0].key) {
				const long gm = i - g0;
				if (gm > 1 && gm > bigm) { big0 = g0; bigm = gm; }
				g0 = i;
			}
		}
		g0 = 0;
		for (long i = 1; i <= m; i++) {
			if (i == m || tmp[i].key != tmp[g0].key) {
				const long gm = i - g0;
				if (gm > 1 && g0 != big0) refine(pk, n, sa0, sa + g0, gm, depth + 16);
				g0 = i;
			}
		}
		if (big0 < 0) break;
		stalled = depth >= 48 && bigm >= STALL_MIN && bigm >= m - (m >> 3);
		sa += big0; m = bigm; depth += 16;
	}
	free(heap);
}

/* sort of (key, idx) pairs with every core: chunks sorted in parallel, then merged pairwise */
static void kv_sort_parallel(kv_t *a, long m, int nthreads) {
	if (m < (1L << 18) || nthreads < 2) { kv_sort(a, m); return; }
	int T = 1;
	while (T * 2 <= nthreads && T < 64) T *= 2;
	long *cut = (long *)malloc(sizeof(long) * (size_t)(T + 1));
	for (int t = 0; t <= T; t++) cut[t] = m * t / T;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
	for (int t = 0; t < T; t++) kv_sort(a + cut[t], cut[t + 1] - cut[t]);
	kv_t *b = kv_alloc(m), *src = a, *dst = b;
	for (int w = 1; w < T; w *= 2) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
		for (int t = 0; t < T; t += 2 * w) {
			long i = cut[t], ie = cut[t + w], j = ie, je = cut[t + 2 * w], o = cut[t];
			while (i < ie && j < je) dst[o++] = (src[j].key < src[i].key) ? src[j++] : src[i++];
			while (i < ie) dst[o++] = src[i++];
			while (j < je) dst[o++] = src[j++];
		}
		kv_t *x = src; src = dst; dst = x;
	}
	if (src != a) memcpy(a, src, sizeof(kv_t) * (size_t)m);
	free(b);
	free(cut);
}

/* Prefix doubling (Larsson-Sadakane style) over the groups refine() handed over.  Every suffix outside them is at its final
 * row; the members of a group occupy their final range of rows and agree on >= h characters, so their order is the order of
 * the suffixes h characters further on, which ISA - the row of a placed suffix, the first row of its group otherwise - answers
 * for both kinds.  Each round doubles h; only rows still in groups are touched.  SA = rows 0..n (SA[0] = n, the empty suffix). */
static void finish_deep_groups(uint64_t *SA, uint64_t n) {
	if (!g_ndeep) return;
	uint64_t h = g_deep[0].depth, tot = 0;
	for (size_t g = 0; g < g_ndeep; g++) { if (g_deep[g].depth < h) h = g_deep[g].depth; tot += g_deep[g].m; }
	printf("Suffix sort: %zu group(s) with %llu suffixes share more than %llu characters; finishing them by prefix doubling\n", g_ndeep, (unsigned long long)tot, (unsigned long long)h);
	uint64_t *ISA = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
	if (!ISA) bwb_die("is_bwt: Could not allocate memory for the BWT index construction, alloc'ing %llu Mb", (unsigned long long)((n + 1) * 8 >> 20));
#pragma omp parallel for schedule(static)
	for (uint64_t r = 0; r <= n; r++) ISA[SA[r]] = r;
	/* deep_t.start counts from SA[1] */
	for (size_t g = 0; g < g_ndeep; g++) g_deep[g].start += 1;
#pragma omp parallel for schedule(dynamic, 64)
	for (size_t g = 0; g < g_ndeep; g++)
		for (uint64_t k = 0; k < g_deep[g].m; k++) ISA[SA[g_deep[g].start + k]] = g_deep[g].start;
	int nthreads = 1;
#pragma omp parallel
#pragma omp single
	nthreads = omp_get_num_threads();
	deep_t *cur = g_deep;
	size_t ncur = g_ndeep;
	g_deep = NULL; g_ndeep = g_capdeep = 0;
	while (ncur) {
		/* phase A: order every group by the rank of the suffix h further on (ISA is only read), remember the keys */
		uint64_t **keys = (uint64_t **)calloc(ncur, sizeof(uint64_t *));
		/* large groups one after the other with all threads, the rest in parallel */
		for (size_t g = 0; g < ncur; g++) {
			if (cur[g].m < (1u << 18)) continue;
			const long m = (long)cur[g].m;
			uint64_t *sa = SA + cur[g].start;
			kv_t *t = kv_alloc(m);
#pragma omp parallel for schedule(static)
			for (long i = 0; i < m; i++) { t[i].key = ISA[sa[i] + h]; t[i].idx = sa[i]; }
			kv_sort_parallel(t, m, nthreads);
			keys[g] = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
			if (!keys[g]) bwb_die("is_bwt: out of memory");
#pragma omp parallel for schedule(static)
			for (long i = 0; i < m; i++) { sa[i] = t[i].idx; keys[g][i] = t[i].key; }
			free(t);
		}
#pragma omp parallel for schedule(dynamic, 8)
		for (size_t g = 0; g < ncur; g++) {
			if (cur[g].m >= (1u << 18)) continue;
			const long m = (long)cur[g].m;
			uint64_t *sa = SA + cur[g].start;
			kv_t *t = kv_alloc(m);
			for (long i = 0; i < m; i++) { t[i].key = ISA[sa[i] + h]; t[i].idx = sa[i]; }
			kv_sort(t, m);
			keys[g] = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m);
			if (!keys[g]) bwb_die("is_bwt: out of memory");
			for (long i = 0; i < m; i++) { sa[i] = t[i].idx; keys[g][i] = t[i].key; }
			free(t);
		}
		/* phase B: split at key changes, give the members their new group row, collect what is still unsorted */
#pragma omp parallel for schedule(dynamic, 8)
		for (size_t g = 0; g < ncur; g++) {
			const uint64_t m = cur[g].m, st = cur[g].start;
			uint64_t g0 = 0;
			for (uint64_t i = 1; i <= m; i++) {
				if (i == m || keys[g][i] != keys[g][g0]) {
					for (uint64_t k = g0; k < i; k++) ISA[SA[st + k]] = st + g0;
					if (i - g0 > 1) deep_add(st + g0, i - g0, 2 * h);
					g0 = i;
				}
			}
			free(keys[g]);
		}
		free(keys);
		free(cur);
		cur = g_deep; ncur = g_ndeep;
		g_deep = NULL; g_ndeep = g_capdeep = 0;
		h *= 2;
	}
	free(cur);
	free(ISA);
}

/* SA of seq[0..n) in the reference's row convention: SA[0] = n (empty suffix), is.c:197-206 */
static uint64_t *build_sa(const unsigned char *seq, uint64_t n) {
	uint64_t *SA = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
	unsigned char *pk = (unsigned char *)calloc((n >> 1) + 64, 1);
	if (!SA || !pk) bwb_die("is_bwt: Could not allocate memory for the BWT index construction, alloc'ing %llu Mb", (unsigned long long)((n + 1) * 8 / 1024 / 1024));
	for (uint64_t i = 0; i < n; i++) pk[i >> 1] |= (unsigned char)(seq[i] << ((i & 1) ? 0 : 4));
	SA[0] = n;
	const int RB = 20; /* 5 characters */
	const uint64_t NB = 1ull << RB;
	uint64_t *cnt = (uint64_t *)calloc(NB + 1, sizeof(uint64_t));
	uint64_t *fill = (uint64_t *)malloc(sizeof(uint64_t) * NB);
	if (!cnt || !fill) bwb_die("is_bwt: out of memory");
	/* bucket id = first 5 characters (zero padded); suffixes shorter than 5 sort first inside their bucket via refine() */
#define BKT(i) (key16(pk, (i)) >> (64 - RB))
	for (uint64_t i = 0; i < n; i++) cnt[BKT(i) + 1]++;
	for (uint64_t b = 0; b < NB; b++) cnt[b + 1] += cnt[b];
	memcpy(fill, cnt, sizeof(uint64_t) * NB);
	uint64_t *S = SA + 1;
	for (uint64_t i = 0; i < n; i++) S[fill[BKT(i)]++] = i;
	free(fill);
	g_ndeep = 0;
#pragma omp parallel for schedule(dynamic, 16)
	for (long b = 0; b < (long)NB; b++) {
		const long m = (long)(cnt[b + 1] - cnt[b]);
		if (m > 1) refine(pk, n, S, S + cnt[b], m, 0);
	}
#undef BKT
	free(cnt);
	free(pk);
	finish_deep_groups(SA, n);
	return SA;
}

/* bwt.c:161-218 (is_bwt is.c:214-243, pack_word io.c:590-609, compute_C bwt.c:266-277, compute_O bwt.c:280-291) */
bwt_t *construct_bwt(unsigned char *ref, bwtint_t length, const char *extSAFname) {
	bwt_t *B = (bwt_t *)calloc(1, sizeof(bwt_t));
	B->length = length + 1;
	B->num_sa = (bwtint_t)ceil(((double)B->length) / SA_INTERVAL);
	B->SA = (bwtint_t *)calloc(B->num_sa, sizeof(bwtint_t));
	unsigned char *bw = (unsigned char *)malloc(B->length);
	if (!B->SA || !bw) bwb_die("construct_bwt: Could not allocate memory for the compressed SA");
	if (extSAFname) {
		/* esa2bwt, bwt.c:132-158: the suffix array comes from an external sorter (eSAIS) as 40-bit little-endian entries for
		 * rows 1..n (row 0 is the empty suffix); it is streamed, never held in memory */
		printf("Computing BWT from precomputed eSAIS SA \n");
		FILE *sf = fopen(extSAFname, "rb");
		if (!sf) bwb_die("esa2bwt: Cannot open the ext SA file: %s!", extSAFname);
		B->SA[0] = length;
		bw[0] = ref[length - 1];
		int have0 = 0;
		for (bwtint_t i = 1; i <= length; i++) {
			bwtint_t v = 0;
			if (fread(&v, 5, 1, sf) < 1) bwb_die("esa2bwt: Could not read ext SA from file: %s!", extSAFname);
			if (v >= length) bwb_die("esa2bwt: suffix %llu in %s is outside the text", (unsigned long long)v, extSAFname);
			if (i % SA_INTERVAL == 0) B->SA[i / SA_INTERVAL] = v;
			if (v == 0) { B->sa0_index = i; bw[i] = 0; have0 = 1; }
			else bw[i] = ref[v - 1];
		}
		fclose(sf);
		if (!have0) bwb_die("esa2bwt: %s does not contain suffix 0", extSAFname);
	} else {
		double t = wall();
		uint64_t *SA = build_sa(ref, length);
		printf("Suffix sort time: %.2f sec\n", wall() - t);
		FILE *dump = getenv("BWB_DUMP_SA") ? fopen(getenv("BWB_DUMP_SA"), "wb") : NULL; /* test aid: the SA in the external 40-bit format */
		for (bwtint_t i = 0; i <= length; i++) {
			if (i % SA_INTERVAL == 0) B->SA[i / SA_INTERVAL] = SA[i];
			if (SA[i] == 0) { B->sa0_index = i; bw[i] = 0; }
			else bw[i] = ref[SA[i] - 1];
			if (dump && i) fwrite(&SA[i], 5, 1, dump);
		}
		if (dump) fclose(dump);
		free(SA);
	}
	B->num_words = (bwtint_t)ceil(((double)B->length) / 8);
	B->bwt = (uint32_t *)calloc(B->num_words, sizeof(uint32_t));
	B->num_occ = (bwtint_t)ceil(((double)B->length) / OCC_INTERVAL);
	B->O = (bwtint_t *)calloc(B->num_occ * ALPHABET_SIZE, sizeof(bwtint_t));
	if (!B->bwt || !B->O) bwb_die("construct_bwt: Could not allocate memory for the BWT index");
	for (bwtint_t i = 0; i < B->length; i++) B->bwt[i >> 3] |= (uint32_t)bw[i] << (28 - 4 * (i & 7)); /* first char in bits 31-28 */
	bwtint_t occ[ALPHABET_SIZE] = { 0 };
	for (bwtint_t i = 0; i < B->length; i++) {
		if (i != B->sa0_index) { B->C[bw[i] + 1]++; occ[bw[i]]++; }
		if (i % OCC_INTERVAL == 0) memcpy(&B->O[(i / OCC_INTERVAL) * ALPHABET_SIZE], occ, sizeof(occ));
	}
	for (int i = 1; i <= ALPHABET_SIZE; i++) B->C[i] += B->C[i - 1];
	free(bw);
	return B;
}

int index_bwt(const char *fastaFname, const char *extSAFname) { /* bwt.c:29-63 */
	printf("**** BWT Index **** \n");
	size_t L = strlen(fastaFname) + 8;
	char *annFname = (char *)malloc(L), *bwtFname = (char *)malloc(L), *refFname = (char *)malloc(L);
	snprintf(annFname, L, "%s.ann", fastaFname);
	snprintf(bwtFname, L, "%s.bwt", fastaFname);
	snprintf(refFname, L, "%s.ref", fastaFname);
	unsigned char *seq;
	bwtint_t seqLen;
	if (!extSAFname) fasta2ref(fastaFname, refFname, annFname, &seq, &seqLen);
	else { /* ref2seq, io.c:158-185: the text comes from the .ref file that an earlier fasta2ref wrote (the external sorter worked on it) */
		FILE *rf = fopen(refFname, "rb");
		if (!rf) bwb_die("ref2seq: Cannot open .ref file: %s!", refFname);
		fseek(rf, 0, SEEK_END);
		long sz = ftell(rf);
		fseek(rf, 0, SEEK_SET);
		seq = (unsigned char *)malloc((size_t)sz + 16);
		if (!seq || fread(seq, 1, (size_t)sz, rf) != (size_t)sz) bwb_die("ref2seq: Could not read the .ref file: %s!", refFname);
		fclose(rf);
		seqLen = (bwtint_t)sz;
		printf("Done reading FASTA file. Total sequence length read = %llu\n", (unsigned long long)seqLen);
	}
	double t = wall();
	bwt_t *B = construct_bwt(seq, seqLen, extSAFname);
	printf("Total BWT construction time: %.2f sec\n", wall() - t);
	free(seq);
	store_bwt(B, bwtFname);
	free(annFname); free(bwtFname); free(refFname);
	free_bwt(B);
	return 0;
}
