/*
 * bwb_oracle - TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the reference's `bwbble align` hot path (viq854/bwbble,
 * mg-aligner/{bwt.c,inexact_match.c,exact_match.c,align.c,io.c}).  It exists so that the HIP
 * path can be checked bit-for-bit on any machine (the reference sources do not travel to the GPU
 * box).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (bwbble_amd/) never links, imports or executes it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against golden
 * vectors produced by the real reference (built by oracle/Makefile into oracle/_ref/bwbble and
 * run by tests/golden/make_golden.py): byte-identical .aln files for -n 0,1,2,3,5 and a gapped
 * configuration, O_alphabet / O known-answer vectors, and per-read D[] arrays.
 */
#ifndef BWB_ORACLE_H
#define BWB_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirror of aln_params_t (mg-aligner/align.h:48-79), same field order */
typedef struct {
	int max_diff, max_gapo, max_gape, max_entries;
	int mm_score, gapo_score, gape_score;
	int seed_length, max_diff_seed, max_best, no_indel_length;
	int matched_Ncontig, use_precalc, is_multiref, n_threads;
} bwb_or_params;

/* mirror of bwt_t (mg-aligner/bwt.h:19-40) without the 64 K LUT */
typedef struct {
	uint64_t length, num_words, num_sa, num_occ, sa0_index;
	uint64_t C[17];
	uint32_t *bwt;
	uint64_t *O;
	uint64_t *SA; /* NULL unless loaded */
} bwb_or_index;

/* mirror of diff_lower_bound_t (mg-aligner/inexact_match.h:11-14) */
typedef struct { int num_diff; int sa_intv_width; } bwb_or_dlb;

/* work counters defined by SURVEY.md 8(d): one "visit" = one (checkpoint row + BWT block) fetch */
typedef struct {
	uint64_t visits_single;   /* groups of <=7 O() calls sharing one position (0 when special-cased) */
	uint64_t visits_alphabet; /* O_alphabet calls with i not in {-1, length-1} */
	uint64_t heap_pops, heap_pushes, max_heap_entries;
	uint64_t n_alignments;
} bwb_or_stats;

void bwb_or_default_params(bwb_or_params *p);                       /* align.c:22-38 */
bwb_or_index *bwb_or_load_bwt(const char *path, int load_sa);       /* bwt.c:90-125 */
void bwb_or_free_index(bwb_or_index *idx);

uint64_t bwb_or_O(const bwb_or_index *idx, int c, uint64_t i);                     /* bwt.c:348-372 */
void bwb_or_O_alphabet(const bwb_or_index *idx, uint64_t i, uint64_t occ[16], int inc); /* bwt.c:374-438 */
uint64_t bwb_or_invPsi(const bwb_or_index *idx, uint64_t i);                       /* bwt.c:311-317 */
uint64_t bwb_or_SA(const bwb_or_index *idx, uint64_t i);                           /* bwt.c:320-329 */

/* vector forms for the known-answer tests */
void bwb_or_O_alphabet_many(const bwb_or_index *idx, const uint64_t *pos, size_t n, int inc, uint64_t *out /* n*16 */);

/* read encoding (io.c:467,502-504; io.h:110,113-130): ASCII -> A0 G1 C2 T3 other 4 */
void bwb_or_encode_read(const char *ascii, int len, uint8_t *seq, uint8_t *rc);

/* inexact_match.c:171-254; D has len+1 entries */
void bwb_or_calculate_d(const bwb_or_index *idx, const uint8_t *seq, int len, bwb_or_dlb *D,
                        const bwb_or_params *p, bwb_or_stats *st);

/* An aligner "thread": owns D, D_seed and the heap exactly as one OpenMP thread of
 * align_reads_inexact_parallel does (inexact_match.c:118-123). D_seed persists across reads. */
typedef struct bwb_or_aligner bwb_or_aligner;
bwb_or_aligner *bwb_or_aligner_new(const bwb_or_index *idx, const bwb_or_params *p, int max_len);
void bwb_or_aligner_free(bwb_or_aligner *a);
/* Aligns one read; appends its .aln record (align.c:345-382) to *buf (realloc'd), returns #entries.
 * fresh_dseed!=0 zeroes D_seed first (models "first read of a thread"). */
int bwb_or_align_read(bwb_or_aligner *a, const uint8_t *seq, const uint8_t *rc, int len, int fresh_dseed,
                      uint8_t **buf, size_t *buf_len, size_t *buf_cap, bwb_or_stats *st);
/* D / D_seed of the last read aligned by this aligner (for known-answer tests) */
const bwb_or_dlb *bwb_or_aligner_D(const bwb_or_aligner *a);
const bwb_or_dlb *bwb_or_aligner_Dseed(const bwb_or_aligner *a);

/* Whole-file driver == align_reads (align.c:40-87) + align_reads_inexact[_parallel]
 * (inexact_match.c:25-168): FASTQ in, .aln out.  n_threads>1 uses the same static chunking.
 * max_reads>0 truncates the input.  Returns #reads, fills stats (summed) and seconds spent in the
 * align loop (wall clock, excluding index/FASTQ load and the .aln write). */
long bwb_or_align_fastq(const char *bwt_path, const char *fastq_path, const char *aln_path,
                        const bwb_or_params *p, int fresh_dseed, long max_reads,
                        bwb_or_stats *st, double *align_seconds);

/* Same, on an already loaded index and already encoded reads (used by bench.py's cpu_baseline) */
long bwb_or_align_encoded(const bwb_or_index *idx, const uint8_t *seqs, const uint16_t *lens, int stride,
                          long n_reads, const bwb_or_params *p, int fresh_dseed,
                          uint8_t **aln_bytes, size_t *aln_len, bwb_or_stats *st, double *align_seconds);
void bwb_or_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
