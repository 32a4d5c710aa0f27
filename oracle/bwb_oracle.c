/*
 * bwb_oracle.c - TEST INFRASTRUCTURE ONLY (see bwb_oracle.h).
 *
 * CPU restatement of the reference hot path.  Written from the reference's observable behaviour;
 * every function cites the reference lines it follows.  Linked lists and the 64 K zero-nibble LUT
 * of the reference are replaced by arrays / popcounts, but every decision that determines the
 * .aln bytes (visit order, merge rule, push order, LIFO pops, break/continue order, 8-bit field
 * wrap, int truncation) is kept.
 */
#define _GNU_SOURCE
#include "bwb_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OCC_INTERVAL 128     /* bwt.h:14 */
#define SA_INTERVAL 32       /* bwt.h:16 */
#define READ_BATCH_SIZE 0x40000 /* align.h:14 */
#define ALN_PATH_ALLOC 256   /* align.h:21 */
#define STATE_M 0
#define STATE_I 1
#define STATE_D 2

/* io.h:29,33,102-110 */
static const unsigned char grayVal[16] = { 0, 1, 3, 2, 6, 7, 5, 4, 12, 13, 15, 14, 10, 11, 9, 8 };
static const unsigned char nucl_bases_table[4][7] = {
	{ 8, 9, 11, 12, 13, 14, 15 }, { 2, 3, 4, 5, 11, 12, 13 }, { 4, 5, 6, 7, 8, 9, 11 }, { 1, 2, 5, 6, 9, 13, 14 } };
static const unsigned char nt4_gray_val[5] = { 8, 2, 4, 1, 15 };

static double now_sec(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void bwb_or_free(void *p) { free(p); }

/* align.c:22-38 */
void bwb_or_default_params(bwb_or_params *p) {
	memset(p, 0, sizeof(*p));
	p->gape_score = 4; p->gapo_score = 11; p->mm_score = 3; p->max_diff = 0; p->max_gape = 6; p->max_gapo = 1;
	p->seed_length = 32; p->max_diff_seed = 2; p->max_entries = 3000000; p->use_precalc = 0;
	p->matched_Ncontig = 0; p->is_multiref = 1; p->max_best = 30; p->no_indel_length = 5; p->n_threads = 1;
}

/* bwt.c:90-125 (file layout bwt.c:66-82) */
bwb_or_index *bwb_or_load_bwt(const char *path, int load_sa) {
	FILE *f = fopen(path, "rb");
	if (!f) return NULL;
	bwb_or_index *x = (bwb_or_index *)calloc(1, sizeof(*x));
	uint64_t hdr[5];
	if (fread(hdr, 8, 5, f) != 5 || fread(x->C, 8, 17, f) != 17) goto bad;
	x->length = hdr[0]; x->num_words = hdr[1]; x->num_sa = hdr[2]; x->num_occ = hdr[3]; x->sa0_index = hdr[4];
	x->bwt = (uint32_t *)calloc(x->num_words + 1, 4);
	x->O = (uint64_t *)calloc(x->num_occ * 16, 8);
	if (!x->bwt || !x->O) goto bad;
	if (fread(x->bwt, 4, x->num_words, f) != x->num_words) goto bad;
	if (fread(x->O, 8, x->num_occ * 16, f) != x->num_occ * 16) goto bad;
	if (load_sa) {
		x->SA = (uint64_t *)calloc(x->num_sa, 8);
		if (!x->SA || fread(x->SA, 8, x->num_sa, f) != x->num_sa) goto bad;
	}
	fclose(f);
	return x;
bad:
	fclose(f);
	bwb_or_free_index(x);
	return NULL;
}

void bwb_or_free_index(bwb_or_index *x) {
	if (!x) return;
	free(x->bwt); free(x->O); free(x->SA); free(x);
}

/* number of 4-bit fields of w equal to c == occ_count_table lookup of (w ^ char_masks[c]),
 * bwt.c:525-536,584-586 */
static inline int nib_eq(uint32_t w, int c) {
	uint32_t x = w ^ (0x11111111u * (uint32_t)c);
	x |= x >> 1; x |= x >> 2; x &= 0x11111111u;
	return 8 - __builtin_popcount(x);
}
/* word_fractions_masks (io.h:94-95): keep the first k chars (k=1..8) of a word, force the rest non-zero */
static inline uint32_t frac_mask(int k) { return k >= 8 ? 0u : (0xFFFFFFFFu >> (4 * k)); }

/* B(), bwt.c:337-345 */
static inline int bwt_char(const bwb_or_index *x, uint64_t i) {
	return (x->bwt[i >> 3] >> (28 - 4 * (i & 7))) & 15;
}

/* O(), bwt.c:348-372 with get_occ_count_opt bwt.c:575-600 */
uint64_t bwb_or_O(const bwb_or_index *x, int c, uint64_t i) {
	if (i == x->length - 1) return x->C[c + 1] - x->C[c];
	if (i == (uint64_t)-1) return 0;
	const uint64_t k = i / OCC_INTERVAL;
	uint64_t o = x->O[k * 16 + c];
	if (c != 0) {
		const uint64_t start = k * OCC_INTERVAL, w0 = start >> 3;
		const uint64_t nw = (i - start + 1) >> 3;
		const int nch = (int)(i - (start + (nw << 3)) + 1);
		uint64_t t = 0;
		for (uint64_t w = w0; w < w0 + nw; w++) t += nib_eq(x->bwt[w], c);
		if (nch > 0) {
			uint32_t w = (x->bwt[w0 + nw] ^ (0x11111111u * (uint32_t)c)) | frac_mask(nch);
			w |= w >> 1; w |= w >> 2; w &= 0x11111111u;
			t += 8 - __builtin_popcount(w);
		}
		if ((int)(x->bwt[w0] >> 28) == c) t--;
		o += t;
	} else {
		for (uint64_t j = k * OCC_INTERVAL + 1; j <= i; j++) {
			if (j == x->sa0_index) continue;
			if (bwt_char(x, j) == 0) o++;
		}
	}
	return o;
}

/* O_alphabet(), bwt.c:374-438 with get_occ_count_alphabet bwt.c:689-781.
 * NB: codes 5,9,11,13 (three-base codes) are NOT counted in the general case (bwt.c:427-435,
 * 718-726), only the "first char of the block" correction bwt.c:780 applies to them. */
void bwb_or_O_alphabet(const bwb_or_index *x, uint64_t i, uint64_t occ[16], int inc) {
	if (i == x->length - 1) { for (int j = 1; j < 16; j++) occ[j] = x->C[j + 1] + inc; return; }
	if (i == (uint64_t)-1) { for (int j = 1; j < 16; j++) occ[j] = x->C[j] + inc; return; }
	const uint64_t k = i / OCC_INTERVAL, s = k * 16;
	const uint64_t start = k * OCC_INTERVAL, w0 = start >> 3;
	const uint64_t nw = (i - start + 1) >> 3;
	const int nch = (int)(i - (start + (nw << 3)) + 1);
	static const int counted[11] = { 1, 2, 3, 4, 6, 7, 8, 10, 12, 14, 15 };
	for (uint64_t w = w0; w < w0 + nw; w++)
		for (int q = 0; q < 11; q++) occ[counted[q]] += nib_eq(x->bwt[w], counted[q]);
	if (nch > 0) {
		const uint32_t wl = x->bwt[w0 + nw], fm = frac_mask(nch);
		for (int q = 0; q < 11; q++) {
			uint32_t w = (wl ^ (0x11111111u * (uint32_t)counted[q])) | fm;
			w |= w >> 1; w |= w >> 2; w &= 0x11111111u;
			occ[counted[q]] += 8 - __builtin_popcount(w);
		}
	}
	occ[x->bwt[w0] >> 28]--; /* bwt.c:780 (any code, including 0 and the uncounted ones) */
	for (int j = 1; j < 16; j++) {
		if (j == 5 || j == 9 || j == 11 || j == 13) occ[j] += x->C[j] + inc;
		else occ[j] += x->C[j] + x->O[s + j] + inc;
	}
}

void bwb_or_O_alphabet_many(const bwb_or_index *x, const uint64_t *pos, size_t n, int inc, uint64_t *out) {
	for (size_t q = 0; q < n; q++) {
		uint64_t occ[16] = { 0 }; /* caller zero-inits, inexact_match.c:377-378 */
		bwb_or_O_alphabet(x, pos[q], occ, inc);
		memcpy(out + 16 * q, occ, sizeof(occ));
	}
}

/* invPsi bwt.c:311-317, SA bwt.c:320-329 */
uint64_t bwb_or_invPsi(const bwb_or_index *x, uint64_t i) {
	if (i == x->sa0_index) return 0;
	int c = bwt_char(x, i);
	return x->C[c] + bwb_or_O(x, c, i);
}
uint64_t bwb_or_SA(const bwb_or_index *x, uint64_t i) {
	uint64_t j = 0;
	while (i % SA_INTERVAL != 0) { i = bwb_or_invPsi(x, i); j++; }
	return (x->SA[i / SA_INTERVAL] + j) % x->length;
}

/* io.c:467 (nt4_table) and io.c:502-504 (nt4_complement) */
void bwb_or_encode_read(const char *ascii, int len, uint8_t *seq, uint8_t *rc) {
	for (int i = 0; i < len; i++) {
		char c = ascii[i];
		seq[i] = (c == 'A' || c == 'a') ? 0 : (c == 'G' || c == 'g') ? 1 : (c == 'C' || c == 'c') ? 2 : (c == 'T' || c == 't') ? 3 : 4;
	}
	for (int i = 0; i < len; i++) rc[len - 1 - i] = seq[i] > 3 ? 4 : 3 - seq[i];
}

/* ---- SA interval list (align.c:93-132): array instead of a linked list, same append/merge rule ---- */
typedef struct { uint64_t L, U; } intv_t;
typedef struct { intv_t *v; int size, cap; } ilist_t;
static void il_add(ilist_t *l, uint64_t L, uint64_t U) { /* align.c:93-110 */
	if (l->size != 0 && L == l->v[l->size - 1].U + 1) { l->v[l->size - 1].U = U; return; }
	if (l->size == l->cap) { l->cap = l->cap ? 2 * l->cap : 16; l->v = (intv_t *)realloc(l->v, sizeof(intv_t) * l->cap); }
	l->v[l->size].L = L; l->v[l->size].U = U; l->size++;
}

/* one multi-interval backward step shared by calculate_d (inexact_match.c:219-232) and
 * exact_match_bounded (exact_match.c:88-109); returns the int-truncated sum of widths */
static int il_step(const bwb_or_index *x, const ilist_t *cur, ilist_t *next, int c, bwb_or_stats *st) {
	int num_matches = 0;
	for (int s = 0; s < cur->size; s++) {
		const uint64_t iL = cur->v[s].L - 1, iU = cur->v[s].U;
		if (st) {
			if (iL != (uint64_t)-1 && iL != x->length - 1) st->visits_single++;
			if (iU != (uint64_t)-1 && iU != x->length - 1) st->visits_single++;
		}
		for (int b = 0; b < 7; b++) {
			const int base = nucl_bases_table[c][b]; /* never N(10) */
			const uint64_t L = x->C[base] + bwb_or_O(x, base, iL) + 1;
			const uint64_t U = x->C[base] + bwb_or_O(x, base, iU);
			if (L <= U) { num_matches += (int)(U - L + 1); il_add(next, L, U); }
		}
	}
	return num_matches;
}

/* calculate_d, multiref branch inexact_match.c:208-253 */
static void calc_d(const bwb_or_index *x, const uint8_t *read, int len, bwb_or_dlb *D, ilist_t *cur, ilist_t *next, bwb_or_stats *st) {
	int z = 0;
	const uint64_t L = 0, U = x->length - 1; /* the outer L,U that inexact_match.c:243 sees */
	cur->size = 0; next->size = 0;
	il_add(cur, L, U);
	for (int i = len - 1; i >= 0; i--) {
		const int c = read[i];
		int num_matches = 0;
		if (c > 3) cur->size = 0;
		else num_matches = il_step(x, cur, next, c, st);
		ilist_t t = *cur; *cur = *next; *next = t;
		next->size = 0;
		if (cur->size == 0) {
			il_add(cur, 0, x->length - 1);
			z++;
			num_matches = (int)(U - L + 1);
		}
		D[len - 1 - i].num_diff = z;
		D[len - 1 - i].sa_intv_width = num_matches;
	}
	D[len].sa_intv_width = 0;
	D[len].num_diff = ++z;
}

/* single-genome branch inexact_match.c:176-206 (codes A=15,G=3,C=7,T=1) */
/* rank-block visits of one (L-1, U) position pair, SURVEY 8(d): 0 for the special-cased positions */
static void count_pair(const bwb_or_index *x, uint64_t iL, uint64_t iU, uint64_t *ctr) {
	if (iL != (uint64_t)-1 && iL != x->length - 1) (*ctr)++;
	if (iU != (uint64_t)-1 && iU != x->length - 1) (*ctr)++;
}
static void calc_d_single(const bwb_or_index *x, const uint8_t *read, int len, bwb_or_dlb *D, bwb_or_stats *st) {
	static const int nt4_gray[5] = { 15, 3, 7, 1, 10 };
	int z = 0;
	uint64_t L = 0, U = x->length - 1;
	for (int i = len - 1; i >= 0; i--) {
		int c = nt4_gray[read[i]];
		if (c == 10) { L = 0; U = x->length - 1; z++; }
		else {
			if (st) count_pair(x, L - 1, U, &st->visits_single);
			uint64_t oL = bwb_or_O(x, c, L - 1), oU = (L - 1 == U) ? oL : bwb_or_O(x, c, U);
			L = x->C[c] + oL + 1; U = x->C[c] + oU;
			if (L > U) { L = 0; U = x->length - 1; z++; }
		}
		D[len - 1 - i].num_diff = z;
		D[len - 1 - i].sa_intv_width = (int)(U - L + 1);
	}
	D[len].sa_intv_width = 0;
	D[len].num_diff = ++z;
}

void bwb_or_calculate_d(const bwb_or_index *x, const uint8_t *seq, int len, bwb_or_dlb *D, const bwb_or_params *p, bwb_or_stats *st) {
	if (!p->is_multiref) { calc_d_single(x, seq, len, D, NULL); return; }
	ilist_t a = { 0 }, b = { 0 };
	calc_d(x, seq, len, D, &a, &b, st);
	free(a.v); free(b.v);
}

/* ---- heap (inexact_match.h:17-34, inexact_match.c:510-610) ---- */
typedef struct { /* aln_entry_t, align.h:100-119: all small fields are 8-bit */
	uint64_t L, U;
	uint8_t num_mm, num_gapo, num_gape, num_snps, score, i, state, aln_length;
	uint8_t path[ALN_PATH_ALLOC];
} entry_t;
typedef struct { int n, cap; entry_t *e; } bucket_t;
typedef struct { int best_score, num_buckets, num_entries; bucket_t *b; } heap_t;

typedef struct { /* aln_t, align.h:81-90 */
	int score; uint64_t L, U; int num_mm, num_gapo, num_gape, aln_length; uint8_t *path;
} aln_t;
typedef struct { int n, cap; aln_t *e; } alns_t;

struct bwb_or_aligner {
	const bwb_or_index *x;
	bwb_or_params p;
	int max_len;
	bwb_or_dlb *D, *Dseed;
	heap_t heap;
	ilist_t la, lb;
	alns_t alns;
};

static inline int aln_score(int m, int o, int e, const bwb_or_params *p) { return m * p->mm_score + o * p->gapo_score + e * p->gape_score; }

static void heap_init(heap_t *h, const bwb_or_params *p) { /* inexact_match.c:510-530 */
	h->num_buckets = aln_score(p->max_diff + 1, p->max_gapo + 1, p->max_gape + 1, p);
	h->b = (bucket_t *)calloc(h->num_buckets, sizeof(bucket_t));
	for (int i = 0; i < h->num_buckets; i++) { h->b[i].cap = 4; h->b[i].e = (entry_t *)calloc(4, sizeof(entry_t)); }
	h->best_score = h->num_buckets; h->num_entries = 0;
}
static void heap_reset(heap_t *h) { /* :540-546 */
	for (int i = 0; i < h->num_buckets; i++) h->b[i].n = 0;
	h->best_score = h->num_buckets; h->num_entries = 0;
}
/* heap_push :548-591.  has_path==0 reproduces the NULL aln_path root push (:281). */
static void heap_push(heap_t *h, int i, uint64_t L, uint64_t U, int mm, int gapo, int gape, int state, int snps,
                      int aln_length, const uint8_t *path, int has_path, const bwb_or_params *p) {
	const int score = aln_score(mm, gapo, gape, p);
	bucket_t *hb = &h->b[score];
	if (hb->n == hb->cap) { hb->cap <<= 1; hb->e = (entry_t *)realloc(hb->e, sizeof(entry_t) * hb->cap); }
	entry_t *e = &hb->e[hb->n];
	e->i = (uint8_t)i; e->score = (uint8_t)score; e->L = L; e->U = U;
	e->num_mm = (uint8_t)mm; e->num_gapo = (uint8_t)gapo; e->num_gape = (uint8_t)gape;
	e->state = (uint8_t)state; e->num_snps = (uint8_t)snps; e->aln_length = 0;
	if (has_path) {
		memset(e->path, 0, ALN_PATH_ALLOC);
		memcpy(e->path, path, (size_t)aln_length);
		e->path[aln_length & 255] = (uint8_t)state; /* aln_length is an 8-bit value already */
		e->aln_length = (uint8_t)(aln_length + 1);
	}
	hb->n++; h->num_entries++;
	if (h->best_score > score) h->best_score = score;
}
static void heap_pop(heap_t *h, entry_t *out) { /* :594-610 */
	bucket_t *hb = &h->b[h->best_score];
	*out = hb->e[hb->n - 1];
	hb->n--; h->num_entries--;
	if (hb->n == 0 && h->num_entries) {
		int i;
		for (i = h->best_score + 1; i < h->num_buckets; i++) if (h->b[i].n != 0) break;
		h->best_score = i;
	} else if (h->num_entries == 0) h->best_score = h->num_buckets;
}

static void add_alignment(alns_t *a, const entry_t *e, uint64_t L, uint64_t U, int score) { /* align.c:271-298 */
	if (e->num_gapo) for (int j = 0; j < a->n; j++) if (a->e[j].L == L && a->e[j].U == U) return;
	if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->e = (aln_t *)realloc(a->e, sizeof(aln_t) * a->cap); }
	aln_t *t = &a->e[a->n++];
	t->num_mm = e->num_mm; t->num_gapo = e->num_gapo; t->num_gape = e->num_gape;
	t->L = L; t->U = U; t->score = score; t->aln_length = e->aln_length;
	t->path = (uint8_t *)malloc(e->aln_length ? e->aln_length : 1);
	memcpy(t->path, e->path, e->aln_length);
}

/* exact_match_bounded, multiref branch exact_match.c:79-118: result left in a->la */
static int exact_bounded(bwb_or_aligner *a, const uint8_t *read, uint64_t l, uint64_t u, int i, bwb_or_stats *st) {
	ilist_t *cur = &a->la, *next = &a->lb;
	cur->size = 0; next->size = 0;
	il_add(cur, l, u);
	for (int r = i; r >= 0; r--) {
		const int c = read[r];
		if (c == 4) { cur->size = 0; break; }
		il_step(a->x, cur, next, c, st);
		ilist_t t = *cur; *cur = *next; *next = t;
		next->size = 0;
		if (cur->size == 0) break;
	}
	return cur->size != 0;
}
/* exact_match_1to1_bounded exact_match.c:196-222 as used by exact_match.c:69-77 (-S) */
static int exact_bounded_single(bwb_or_aligner *a, const uint8_t *read, uint64_t l, uint64_t u, int i, bwb_or_stats *st) {
	static const int nt4_gray[5] = { 15, 3, 7, 1, 10 };
	const bwb_or_index *x = a->x;
	a->la.size = 0;
	uint64_t L = l, U = u;
	for (int r = i; r >= 0; r--) {
		if (read[r] > 3) return 0;
		int c = nt4_gray[read[r]];
		if (st) count_pair(x, L - 1, U, &st->visits_single);
		uint64_t oL = bwb_or_O(x, c, L - 1), oU = bwb_or_O(x, c, U);
		L = x->C[c] + oL + 1; U = x->C[c] + oU;
		if (L > U) return 0;
	}
	il_add(&a->la, L, U);
	return 1;
}

/* O_actg_alphabet bwt.c:440-463 (+ :647-687): occ[1..4] = A(15),G(3),C(7),T(1) */
static void O_actg(const bwb_or_index *x, uint64_t i, uint64_t occ[16], int inc) {
	static const int code[5] = { 0, 15, 3, 7, 1 };
	if (i == x->length - 1) { for (int j = 1; j <= 4; j++) occ[j] = x->C[code[j] + 1] + inc; return; }
	if (i == (uint64_t)-1) { for (int j = 1; j <= 4; j++) occ[j] = x->C[code[j]] + inc; return; }
	for (int j = 1; j <= 4; j++) occ[j] = x->C[code[j]] + bwb_or_O(x, code[j], i) + inc;
}

#define BWB_OR_PRECALC_LEN 12 /* PRECALC_INTERVAL_LENGTH align.h:31 */
/* inexact_match, inexact_match.c:256-506 */
static void inexact_match(bwb_or_aligner *a, const uint8_t *read, int readLen, bwb_or_stats *st) {
	const bwb_or_index *x = a->x;
	const bwb_or_params *p = &a->p;
	heap_t *heap = &a->heap;
	alns_t *alns = &a->alns;
	const bwb_or_dlb *D = a->D, *D_seed = a->Dseed;
	static const unsigned char is_snp[16] = { 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0 };

	int countN = 0;
	for (int i = 0; i < readLen; i++) if (read[i] > 3) countN++;
	if (countN > p->max_diff) return;

	heap_reset(heap);
	if (p->use_precalc) {
		/* inexact_match.c:269-279: one entry per interval of the 12-mer's precalculated list, at i = readLen - 12 with a
		 * 12-long all-M path.  The table entry is exact_match() of the 12-mer (align.c:212-216 -> exact_match.c:57-59),
		 * a pure function of it, so it is computed here instead of being read from the .pre file; those rank calls are
		 * not reference work and are not counted. */
		const uint8_t *mer = read + readLen - BWB_OR_PRECALC_LEN;
		int found = p->is_multiref ? exact_bounded(a, mer, 0, x->length - 1, BWB_OR_PRECALC_LEN - 1, NULL)
		                           : exact_bounded_single(a, mer, 0, x->length - 1, BWB_OR_PRECALC_LEN - 1, NULL);
		if (!found) return; /* :277 */
		uint8_t zeros[BWB_OR_PRECALC_LEN] = { 0 };
		for (int k = 0; k < a->la.size; k++) {
			heap_push(heap, readLen - BWB_OR_PRECALC_LEN, a->la.v[k].L, a->la.v[k].U, 0, 0, 0, 0, 0, BWB_OR_PRECALC_LEN - 1, zeros, 1, p);
			if (st) st->heap_pushes++;
		}
	} else {
		heap_push(heap, readLen, 0, x->length - 1, 0, 0, 0, 0, 0, 0, NULL, 0, p);
		if (st) st->heap_pushes++;
	}

	int best_score = aln_score(p->max_diff + 1, p->max_gapo + 1, p->max_gape + 1, p);
	int best_diff = p->max_diff + 1;
	int max_diff = p->max_diff;
	int num_best = 0;
	(void)best_diff;

	while (heap->num_entries != 0) {
		if (st && (uint64_t)heap->num_entries > st->max_heap_entries) st->max_heap_entries = heap->num_entries;
		if (heap->num_entries > p->max_entries) break;
		entry_t e_;
		heap_pop(heap, &e_);
		entry_t *e = &e_;
		if (st) st->heap_pops++;

		if (e->score > best_score + p->mm_score) break;
		int diff_left = max_diff - e->num_mm - e->num_gapo - e->num_gape;
		if (diff_left < 0) continue;
		if (e->i > 0 && diff_left < D[e->i - 1].num_diff) continue;
		int diff_left_seed = p->max_diff_seed - e->num_mm - e->num_gapo - e->num_gape;
		int seed_index = e->i - (readLen - p->seed_length);
		if (seed_index > 0 && diff_left_seed < D_seed[seed_index - 1].num_diff) continue;

		if (e->i == 0) {
			int score = aln_score(e->num_mm, e->num_gapo, e->num_gape, p);
			if (alns->n == 0) {
				best_score = score;
				best_diff = e->num_mm + e->num_gapo + e->num_gape;
				max_diff = (best_diff + 1 > p->max_diff) ? p->max_diff : best_diff + 1;
			}
			if (score == best_score) num_best += (int)(e->U - e->L + 1);
			else if (num_best > p->max_best) break;
			add_alignment(alns, e, e->L, e->U, score);
			continue;
		} else if (diff_left == 0) {
			int found = p->is_multiref ? exact_bounded(a, read, e->L, e->U, e->i - 1, st)
			                           : exact_bounded_single(a, read, e->L, e->U, e->i - 1, st);
			if (found) {
				ilist_t *sa = &a->la;
				int score = aln_score(e->num_mm, e->num_gapo, e->num_gape, p);
				if (alns->n == 0) {
					best_score = score;
					best_diff = e->num_mm + e->num_gapo + e->num_gape;
					max_diff = (best_diff + 1 > p->max_diff) ? p->max_diff : best_diff + 1;
				}
				if (score == best_score) { for (int k = 0; k < sa->size; k++) num_best += (int)(sa->v[k].U - sa->v[k].L + 1); }
				else if (num_best > p->max_best) break;
				/* inexact_match.c:364-365: remaining chars are matches; the path bytes are already 0 = STATE_M
				 * because heap_push memsets the whole path (:579).  The NULL-path root (:281) never had its
				 * path cleared, but bucket 0 only ever holds all-M paths, so it reads zeros too. */
				int old_len = e->aln_length;
				e->aln_length = (uint8_t)(e->aln_length + e->i);
				for (int q = old_len; q < old_len + e->i && q < ALN_PATH_ALLOC; q++) e->path[q] = STATE_M;
				if (old_len == 0) memset(e->path, 0, ALN_PATH_ALLOC);
				for (int k = 0; k < sa->size; k++) add_alignment(alns, e, sa->v[k].L, sa->v[k].U, score);
			}
			continue;
		}

		uint64_t L[16] = { 0 }, U[16] = { 0 };
		int alphabet_size = 16, is_multiref = p->is_multiref;
		if (is_multiref) {
			bwb_or_O_alphabet(x, e->L - 1, L, 1);
			bwb_or_O_alphabet(x, e->U, U, 0);
			if (st) {
				if (e->L - 1 != (uint64_t)-1 && e->L - 1 != x->length - 1) st->visits_alphabet++;
				if (e->U != (uint64_t)-1 && e->U != x->length - 1) st->visits_alphabet++;
			}
		} else {
			O_actg(x, e->L - 1, L, 1);
			O_actg(x, e->U, U, 0);
			if (st) count_pair(x, e->L - 1, e->U, &st->visits_alphabet);
			alphabet_size = 5;
		}

		int allow_diff = 1, allow_indels = 1, allow_mm = 1, allow_open = 1, allow_extend = 1;
		if (e->i - 1 > 0) {
			if ((diff_left - 1) < D[e->i - 2].num_diff) allow_diff = 0;
			else if (D[e->i - 1].num_diff == diff_left - 1 && D[e->i - 2].num_diff == diff_left - 1
			         && D[e->i - 1].sa_intv_width == D[e->i - 2].sa_intv_width) allow_mm = 0;
		}
		if (seed_index - 1 > 0) {
			if ((diff_left_seed - 1) < D_seed[seed_index - 2].num_diff) allow_diff = 0;
			else if (D_seed[seed_index - 1].num_diff == diff_left_seed - 1 && D_seed[seed_index - 2].num_diff == diff_left_seed - 1
			         && D_seed[seed_index - 1].sa_intv_width == D_seed[seed_index - 2].sa_intv_width) allow_mm = 0;
		}
		int tmp = e->num_gapo + e->num_gape;
		if ((e->i - 1 < (p->no_indel_length + tmp)) || ((readLen - (e->i - 1)) < (p->no_indel_length + tmp))) allow_indels = 0;
		if (e->num_gapo >= p->max_gapo && e->num_gape >= p->max_gape) allow_indels = 0;
		if (e->num_gapo >= p->max_gapo) allow_open = 0;
		if (e->num_gape >= p->max_gape) allow_extend = 0;

		const int before = heap->num_entries;
		if (allow_diff && allow_indels) {
			if (e->state == STATE_I) {
				if (allow_extend)
					heap_push(heap, e->i - 1, e->L, e->U, e->num_mm, e->num_gapo, e->num_gape + 1, STATE_I, e->num_snps, e->aln_length, e->path, 1, p);
			} else {
				if (allow_open && e->state == STATE_M)
					heap_push(heap, e->i - 1, e->L, e->U, e->num_mm, e->num_gapo + 1, e->num_gape, STATE_I, e->num_snps, e->aln_length, e->path, 1, p);
				for (int j = 1; j < alphabet_size; j++) {
					if (L[j] <= U[j]) {
						if (e->state == STATE_M) {
							if (allow_open)
								heap_push(heap, e->i, L[j], U[j], e->num_mm, e->num_gapo + 1, e->num_gape, STATE_D, e->num_snps, e->aln_length, e->path, 1, p);
						} else if (allow_extend)
							heap_push(heap, e->i, L[j], U[j], e->num_mm, e->num_gapo, e->num_gape + 1, STATE_D, e->num_snps, e->aln_length, e->path, 1, p);
					}
				}
			}
		}
		const int c = read[e->i - 1];
		if (allow_diff && allow_mm) {
			for (int j = 1; j < alphabet_size; j++) {
				if (L[j] <= U[j]) {
					int is_mm = 0;
					if (is_multiref) { if (c > 3 || j == 10 || (nt4_gray_val[c] & grayVal[j]) == 0) is_mm = 1; }
					else if (c > 3 || c != j - 1) is_mm = 1;
					heap_push(heap, e->i - 1, L[j], U[j], e->num_mm + is_mm, e->num_gapo, e->num_gape, STATE_M,
					          e->num_snps + (is_multiref && is_snp[j]), e->aln_length, e->path, 1, p);
				}
			}
		} else if (c < 4) {
			if (is_multiref) {
				for (int b = 0; b < 7; b++) {
					int base = nucl_bases_table[c][b];
					if (L[base] <= U[base])
						heap_push(heap, e->i - 1, L[base], U[base], e->num_mm, e->num_gapo, e->num_gape, STATE_M,
						          e->num_snps + is_snp[base], e->aln_length, e->path, 1, p);
				}
			} else if (L[c + 1] <= U[c + 1])
				heap_push(heap, e->i - 1, L[c + 1], U[c + 1], e->num_mm, e->num_gapo, e->num_gape, STATE_M, e->num_snps, e->aln_length, e->path, 1, p);
		}
		if (st) st->heap_pushes += (uint64_t)(heap->num_entries - before);
	}
}

bwb_or_aligner *bwb_or_aligner_new(const bwb_or_index *x, const bwb_or_params *p, int max_len) {
	bwb_or_aligner *a = (bwb_or_aligner *)calloc(1, sizeof(*a));
	a->x = x; a->p = *p; a->max_len = max_len;
	a->D = (bwb_or_dlb *)calloc(max_len + 1, sizeof(bwb_or_dlb));           /* inexact_match.c:34,119 */
	a->Dseed = (bwb_or_dlb *)calloc((p->seed_length > max_len ? p->seed_length : max_len) + 2, sizeof(bwb_or_dlb)); /* :36,121 (padded) */
	heap_init(&a->heap, p);
	return a;
}
void bwb_or_aligner_free(bwb_or_aligner *a) {
	if (!a) return;
	for (int i = 0; i < a->heap.num_buckets; i++) free(a->heap.b[i].e);
	free(a->heap.b); free(a->D); free(a->Dseed); free(a->la.v); free(a->lb.v); free(a->alns.e); free(a);
}
const bwb_or_dlb *bwb_or_aligner_D(const bwb_or_aligner *a) { return a->D; }
const bwb_or_dlb *bwb_or_aligner_Dseed(const bwb_or_aligner *a) { return a->Dseed; }

static void buf_put(uint8_t **buf, size_t *len, size_t *cap, const void *src, size_t n) {
	if (*len + n > *cap) { *cap = (*cap ? *cap * 2 : 4096) + n; *buf = (uint8_t *)realloc(*buf, *cap); }
	memcpy(*buf + *len, src, n); *len += n;
}

/* alns2alnf_bin, align.c:345-382 */
static void alns_to_bytes(const alns_t *a, uint8_t **buf, size_t *len, size_t *cap) {
	int32_t n = a->n;
	buf_put(buf, len, cap, &n, 4);
	for (int i = 0; i < a->n; i++) {
		const aln_t *t = &a->e[i];
		int32_t v;
		v = t->score; buf_put(buf, len, cap, &v, 4);
		buf_put(buf, len, cap, &t->L, 8); buf_put(buf, len, cap, &t->U, 8);
		v = t->num_mm; buf_put(buf, len, cap, &v, 4);
		v = t->num_gapo; buf_put(buf, len, cap, &v, 4);
		v = t->num_gape; buf_put(buf, len, cap, &v, 4);
		v = t->aln_length; buf_put(buf, len, cap, &v, 4);
		int32_t pairs = 0;
		if (t->aln_length > 0) {
			int32_t states[ALN_PATH_ALLOC];
			int state = t->path[t->aln_length - 1];
			uint16_t counter = 1;
			pairs = 1;
			for (int j = t->aln_length - 2; j >= 0; j--) {
				if (state == t->path[j]) counter++;
				else { states[pairs - 1] = state | (counter << 2); state = t->path[j]; counter = 1; pairs++; }
			}
			states[pairs - 1] = state | (counter << 2);
			buf_put(buf, len, cap, &pairs, 4);
			buf_put(buf, len, cap, states, 4 * (size_t)pairs);
		} else buf_put(buf, len, cap, &pairs, 4);
	}
}

int bwb_or_align_read(bwb_or_aligner *a, const uint8_t *seq, const uint8_t *rc, int len, int fresh_dseed,
                      uint8_t **buf, size_t *buf_len, size_t *buf_cap, bwb_or_stats *st) {
	const bwb_or_params *p = &a->p;
	for (int i = 0; i < a->alns.n; i++) free(a->alns.e[i].path);
	a->alns.n = 0;
	if (fresh_dseed) memset(a->Dseed, 0, sizeof(bwb_or_dlb) * ((p->seed_length > a->max_len ? p->seed_length : a->max_len) + 2));
	if (p->use_precalc) { /* inexact_match.c:50-57 / 129-136: reads with an N in the last 12 bases of rc get an empty record */
		/* (len < 12: read2index, align.c:174-186, reads before the buffer in the reference; the product refuses such a batch,
		 * here the read gets an empty record) */
		if (len < BWB_OR_PRECALC_LEN) { alns_to_bytes(&a->alns, buf, buf_len, buf_cap); return 0; }
		for (int i = len - BWB_OR_PRECALC_LEN; i < len; i++)
			if (rc[i] > 3) { alns_to_bytes(&a->alns, buf, buf_len, buf_cap); return 0; }
	}
	/* inexact_match.c:61-65 / 140-144 */
	if (p->is_multiref) calc_d(a->x, seq, len, a->D, &a->la, &a->lb, st); else calc_d_single(a->x, seq, len, a->D, st);
	if (p->seed_length && len > p->seed_length) {
		if (p->is_multiref) calc_d(a->x, seq, p->seed_length, a->Dseed, &a->la, &a->lb, st); else calc_d_single(a->x, seq, p->seed_length, a->Dseed, st);
	}
	inexact_match(a, rc, len, st);
	if (st) st->n_alignments += a->alns.n;
	alns_to_bytes(&a->alns, buf, buf_len, buf_cap);
	return a->alns.n;
}

static void stats_add(bwb_or_stats *d, const bwb_or_stats *s) {
	d->visits_single += s->visits_single; d->visits_alphabet += s->visits_alphabet;
	d->heap_pops += s->heap_pops; d->heap_pushes += s->heap_pushes; d->n_alignments += s->n_alignments;
	if (s->max_heap_entries > d->max_heap_entries) d->max_heap_entries = s->max_heap_entries;
}

long bwb_or_align_encoded(const bwb_or_index *x, const uint8_t *seqs, const uint16_t *lens, int stride, long n_reads,
                          const bwb_or_params *p, int fresh_dseed, uint8_t **aln_bytes, size_t *aln_len,
                          bwb_or_stats *st, double *align_seconds) {
	int max_len = 0;
	for (long i = 0; i < n_reads; i++) if (lens[i] > max_len) max_len = lens[i];
	int T = p->n_threads > 1 ? p->n_threads : 1;
	uint8_t **rbuf = (uint8_t **)calloc(n_reads ? n_reads : 1, sizeof(uint8_t *));
	size_t *rlen = (size_t *)calloc(n_reads ? n_reads : 1, sizeof(size_t));
	bwb_or_stats total; memset(&total, 0, sizeof(total));
	double t_align = 0;
	bwb_or_aligner *seq_aligner = T == 1 ? bwb_or_aligner_new(x, p, max_len) : NULL; /* inexact_match.c:34-38 */
	for (long done = 0; done < n_reads;) {
		long batch = n_reads - done > READ_BATCH_SIZE ? READ_BATCH_SIZE : n_reads - done; /* :44,105 */
		double t0 = now_sec();
#ifdef _OPENMP
		omp_set_num_threads(T);
#endif
#pragma omp parallel if (T > 1)
		{
			int tid = 0, nt = 1;
#ifdef _OPENMP
			if (T > 1) { tid = omp_get_thread_num(); nt = omp_get_num_threads(); }
#endif
			long cs = tid * batch / nt, ce = (tid + 1) * batch / nt; /* :115-116 */
			bwb_or_aligner *a = T == 1 ? seq_aligner : bwb_or_aligner_new(x, p, max_len); /* :119-123 */
			bwb_or_stats lst; memset(&lst, 0, sizeof(lst));
			uint8_t *rc = (uint8_t *)malloc(max_len + 1);
			for (long i = done + cs; i < done + ce; i++) {
				const uint8_t *seq = seqs + (size_t)i * stride;
				int len = lens[i];
				for (int k = 0; k < len; k++) rc[len - 1 - k] = seq[k] > 3 ? 4 : 3 - seq[k];
				size_t cap = 0;
				bwb_or_align_read(a, seq, rc, len, fresh_dseed, &rbuf[i], &rlen[i], &cap, &lst);
			}
			free(rc);
			if (T > 1) bwb_or_aligner_free(a);
#pragma omp critical
			stats_add(&total, &lst);
		}
		t_align += now_sec() - t0;
		done += batch;
	}
	if (seq_aligner) bwb_or_aligner_free(seq_aligner);
	size_t tot = 0;
	for (long i = 0; i < n_reads; i++) tot += rlen[i];
	uint8_t *out = (uint8_t *)malloc(tot ? tot : 1);
	size_t off = 0;
	for (long i = 0; i < n_reads; i++) { memcpy(out + off, rbuf[i], rlen[i]); off += rlen[i]; free(rbuf[i]); }
	free(rbuf); free(rlen);
	*aln_bytes = out; *aln_len = tot;
	if (st) *st = total;
	if (align_seconds) *align_seconds = t_align;
	return n_reads;
}

/* fastq2reads, io.c:410-515: only the sequence line matters for .aln */
static long load_fastq(const char *path, long max_reads, uint8_t **seqs_out, uint16_t **lens_out, int *stride_out) {
	FILE *f = fopen(path, "r");
	if (!f) return -1;
	fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
	char *raw = (char *)malloc(sz + 1);
	if (fread(raw, 1, sz, f) != (size_t)sz) { fclose(f); free(raw); return -1; }
	fclose(f);
	long n = 0, cap = 1024; int stride = 0;
	long *soff = (long *)malloc(sizeof(long) * cap); int *slen = (int *)malloc(sizeof(int) * cap);
	long p = 0;
	while (p < sz && (max_reads <= 0 || n < max_reads)) {
		while (p < sz && raw[p] != '@') p++;              /* io.c:430-434 */
		if (p >= sz) break;
		while (p < sz && raw[p] != '\n') p++;             /* name line */
		p++;
		long s = p;
		while (p < sz && raw[p] != '\n') p++;             /* sequence line */
		int len = (int)(p - s);
		while (p < sz && raw[p] != '+') p++;              /* io.c:474-476 */
		while (p < sz && raw[p] != '\n') p++;
		p++;
		long q = p;
		while (p < sz && raw[p] != '\n') p++;             /* quality line */
		if (p - q != len) { fprintf(stderr, "oracle: quality/sequence length mismatch\n"); exit(1); } /* io.c:495-498 */
		if (n == cap) { cap *= 2; soff = (long *)realloc(soff, sizeof(long) * cap); slen = (int *)realloc(slen, sizeof(int) * cap); }
		soff[n] = s; slen[n] = len; n++;
		if (len > stride) stride = len;
	}
	if (stride == 0) stride = 1;
	uint8_t *seqs = (uint8_t *)calloc((size_t)(n ? n : 1) * stride, 1);
	uint16_t *lens = (uint16_t *)calloc(n ? n : 1, 2);
	uint8_t *tmp = (uint8_t *)malloc(stride + 1);
	for (long i = 0; i < n; i++) {
		bwb_or_encode_read(raw + soff[i], slen[i], seqs + (size_t)i * stride, tmp);
		lens[i] = (uint16_t)slen[i];
	}
	free(tmp); free(raw); free(soff); free(slen);
	*seqs_out = seqs; *lens_out = lens; *stride_out = stride;
	return n;
}

long bwb_or_align_fastq(const char *bwt_path, const char *fastq_path, const char *aln_path, const bwb_or_params *p,
                        int fresh_dseed, long max_reads, bwb_or_stats *st, double *align_seconds) {
	bwb_or_index *x = bwb_or_load_bwt(bwt_path, 0);
	if (!x) return -1;
	uint8_t *seqs; uint16_t *lens; int stride;
	long n = load_fastq(fastq_path, max_reads, &seqs, &lens, &stride);
	if (n < 0) { bwb_or_free_index(x); return -2; }
	uint8_t *bytes; size_t blen;
	bwb_or_align_encoded(x, seqs, lens, stride, n, p, fresh_dseed, &bytes, &blen, st, align_seconds);
	FILE *f = fopen(aln_path, "wb");
	if (!f) { n = -3; }
	else { fwrite(bytes, 1, blen, f); fclose(f); }
	free(bytes); free(seqs); free(lens);
	bwb_or_free_index(x);
	return n;
}

#ifdef BWB_ORACLE_MAIN
/* tiny CLI: bwb_oracle <fasta-prefix> <fastq> <out.aln> [-n N ...same flags as bwbble align] */
#include <unistd.h>
int main(int argc, char **argv) {
	bwb_or_params p; bwb_or_default_params(&p);
	int c, fresh = 0;
	while ((c = getopt(argc, argv, "M:O:E:n:k:o:e:l:m:t:SPF")) >= 0) {
		switch (c) {
		case 'M': p.mm_score = atoi(optarg); break; case 'O': p.gapo_score = atoi(optarg); break;
		case 'E': p.gape_score = atoi(optarg); break; case 'n': p.max_diff = atoi(optarg); break;
		case 'k': p.max_diff_seed = atoi(optarg); break; case 'o': p.max_gapo = atoi(optarg); break;
		case 'e': p.max_gape = atoi(optarg); break; case 'l': p.seed_length = atoi(optarg); break;
		case 'm': p.max_entries = atoi(optarg); break; case 't': p.n_threads = atoi(optarg); break;
		case 'S': p.is_multiref = 0; break; case 'P': p.use_precalc = 1; break; case 'F': fresh = 1; break;
		default: return 1;
		}
	}
	if (argc - optind < 3) { fprintf(stderr, "usage: bwb_oracle [opts] <fasta> <fastq> <out.aln>\n"); return 1; }
	char bwt[4096]; snprintf(bwt, sizeof bwt, "%s.bwt", argv[optind]);
	bwb_or_stats st; double sec;
	long n = bwb_or_align_fastq(bwt, argv[optind + 1], argv[optind + 2], &p, fresh, 0, &st, &sec);
	if (n < 0) { fprintf(stderr, "oracle failed (%ld)\n", n); return 1; }
	printf("reads %ld  align_sec %.3f  reads/s %.1f  visits_single %llu visits_alphabet %llu pops %llu pushes %llu max_heap %llu alns %llu\n",
	       n, sec, n / sec, (unsigned long long)st.visits_single, (unsigned long long)st.visits_alphabet,
	       (unsigned long long)st.heap_pops, (unsigned long long)st.heap_pushes, (unsigned long long)st.max_heap_entries,
	       (unsigned long long)st.n_alignments);
	return 0;
}
#endif
