/*
 * bwb_host.h - host side (C) of the MI355X BWBBLE aligner: the reference's own data types and
 * entry points for the `index | align | aln2sam` path, re-implemented from scratch, with the
 * alignment loop handed to libbwbble_hip.so through include/bwbble_hip.h.
 *
 * Names follow the reference (mg-aligner/{bwt.h,io.h,align.h,inexact_match.h}) so that the seam is
 * obvious: align_reads_inexact_gpu_stream() stands where align_reads_inexact_parallel() (inexact_match.h:40) is
 * selected in align_reads() (align.c:72-76); it takes the FASTQ's NAME instead of the loaded reads_t - the file is streamed
 * through a reader thread while the index is still on its way to the GPU (align_gpu.c).  The drop-in binding with the
 * reference's exact signature, for a maintainer who keeps the reference's host code, is integration/align_gpu.c.
 */
#ifndef BWB_HOST_H
#define BWB_HOST_H
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>
#include <stdio.h>
#include "bwbble_hip.h"

typedef uint64_t bwtint_t;

#define OCC_INTERVAL 128        /* bwt.h:14 */
#define SA_INTERVAL 32          /* bwt.h:16 */
#define ALPHABET_SIZE 16        /* io.h:27 */
#define MAX_SEQ_NAME_LEN 256    /* io.h:10 */

/* bwt_t (bwt.h:19-40) minus the 64 K-entry LUT (popcounts are used instead) */
typedef struct {
	bwtint_t length, num_words;
	uint32_t *bwt;
	bwtint_t C[ALPHABET_SIZE + 1];
	bwtint_t *O;
	bwtint_t num_occ;
	bwtint_t *SA;
	bwtint_t num_sa;
	bwtint_t sa0_index;
	/* load_bwt_start: the arrays are filled by loader threads while the GPUs already take the index in */
	volatile uint64_t blocks_ready;   /* leading 128-character blocks whose bwt words and O rows are in memory (release/acquire) */
	void *loader;                     /* opaque: the loader threads (NULL: everything is loaded) */
	double load_seconds;              /* load_bwt_start to the last byte of the file in memory */
} bwt_t;

/* reads_t/read_t (io.h:151-194) in structure-of-arrays form: codes are read->seq (A0 G1 C2 T3 N4) */
typedef struct {
	unsigned int count, max_len;
	uint32_t stride;            /* bytes per read in seq */
	uint8_t *seq;               /* [count][stride] */
	uint16_t *len;
	char *raw;                  /* the FASTQ text; names and qualities point into it */
	size_t *name_off, *qual_off;
	uint16_t *name_len;
} reads_t;

typedef struct {                /* seq_annotation_t / fasta_annotations_t (io.h:196-207) */
	char name[MAX_SEQ_NAME_LEN + 1];
	bwtint_t start_index, end_index;
} seq_annotation_t;
typedef struct { int num_seq; seq_annotation_t *seq_anns; } fasta_annotations_t;

typedef bwb_params aln_params_t; /* align.h:48-79 */

/* alignments of a batch as loaded from / written to .aln (align.c:345-382,430-483) */
typedef struct {
	size_t n_reads;
	uint64_t *aln_off;          /* n_reads + 1 */
	bwb_aln *alns;
} alns_batch_t;

void bwb_die(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));

/* bwt_io.c */
void store_bwt(const bwt_t *BWT, const char *bwtFname);                /* bwt.c:66-82 */
bwt_t *load_bwt(const char *bwtFname, int loadSA);                     /* bwt.c:90-125 */
double bwt_load_seconds(const bwt_t *B);
bwt_t *load_bwt_start(const char *bwtFname, int loadSA);               /* returns after the header: the arrays fill in the background (blocks_ready) */
void load_bwt_wait(bwt_t *BWT);                                        /* until the whole file is in memory */
void free_bwt(bwt_t *BWT);

/* index.c */
int index_bwt(const char *fastaFname, const char *extSAFname);         /* bwt.c:29-63 */
void fasta2ref(const char *fastaFname, const char *refFname, const char *annFname, unsigned char **seq, bwtint_t *totalSeqLen); /* io.c:190-321 */
bwt_t *construct_bwt(unsigned char *seq, bwtint_t length, const char *extSAFname); /* bwt.c:161-218; extSAFname: esa2bwt bwt.c:132-158 */
fasta_annotations_t *annf2ann(const char *annFname);                   /* io.c:324-349 */
void free_ann(fasta_annotations_t *a);

/* reads.c */
reads_t *fastq2reads(const char *readsFname);                          /* io.c:410-515 */
void free_reads(reads_t *reads);
/* the same parser as a stream: `align` takes the FASTQ in chunks, chunk k+1 is parsed while chunk k is on the GPUs */
typedef struct fq_stream fq_stream;
typedef struct {
	uint32_t n, stride, max_len;  /* reads in the chunk; bytes per read in seq (the chunk's longest read, at least 1) */
	uint8_t *seq;                 /* [n][stride] read->seq codes (A0 G1 C2 T3 N4), malloc'ed: the consumer frees it */
	uint16_t *len;                /* [n], malloc'ed */
} fq_chunk_t;
fq_stream *fq_open(const char *readsFname);
int fq_next_chunk(fq_stream *s, uint32_t max_reads, fq_chunk_t *out); /* 0 at the end of the file */
void fq_close(fq_stream *s);

/* aln_io.c */
void alns2alnf_bin(const bwb_aln *alns, uint64_t n, FILE *alnFile);    /* align.c:345-382, one read */
unsigned char *alns2alnf_buf(const bwb_aln *alns, const uint64_t *aln_off, uint32_t n_reads, size_t *len); /* the same bytes for a chunk of reads, as one buffer */
alns_batch_t *alnsf2alns_bin(const char *alnFname);                    /* align.c:430-483 */
void free_alns_batch(alns_batch_t *b);
int aln_path_bytes(const bwb_aln *a, unsigned char *path /* >= 272 bytes */); /* edit path from the gap runs; returns aln_length */

/* align_gpu.c */
void set_default_aln_params(aln_params_t *params);                     /* align.c:22-38 */
int align_reads(char *fastaFname, char *readsFname, char *alnsFname, aln_params_t *params, int n_gpus); /* align.c:40-87 */
int align_reads_inexact_gpu_stream(bwt_t *BWT, const char *readsFname, aln_params_t *params, char *alnFname, int n_gpus); /* the GPU's align_reads_inexact_parallel (inexact_match.h:40) over a streamed FASTQ */

/* precalc.c */
void precalc_sa_intervals(bwt_t *BWT, const aln_params_t *params, const char *preFname); /* align.c:200-224: writes <fasta>.pre */
int check_precalc_file(const char *preFname); /* the walk of load_precalc_sa_intervals (align.c:226-238) over an existing table: 0 = complete */

/* sam.c */
void alns2sam(char *fastaFname, char *readsFname, char *alnsFname, char *samFname, int is_multiref, int max_diff, int n_gpus); /* align.c:494-556 */


/* The OpenMP teams of the host stages of `align` / `aln2sam` (base encoding, .aln serialisation, SAM text) have at most 32 threads unless
 * OMP_NUM_THREADS says otherwise: on the GPU box (2 x 64 cores, 256 hardware threads) a team of 256 - every hardware thread spinning in
 * libgomp's barriers next to its sibling - parsed 8.8 M reads/s and serialised 8.5 M records/s where 32 threads do 33 M and 240 M
 * (tools/r5_hostbox.sh, profiles/r5_host_stage_threads.txt), and `align -g 8` runs one serialising team per GPU worker.  A clause on
 * every region, not omp_set_num_threads(): the regions run in pthreads, which do not inherit the caller's setting.  `index` keeps the
 * default team. */
static inline int bwb_host_team(void) {
	const char *e = getenv("OMP_NUM_THREADS");
	if (e && atoi(e) > 0) return atoi(e);
	const long nc = sysconf(_SC_NPROCESSORS_ONLN);
	return nc < 1 ? 1 : (nc < 32 ? (int)nc : 32);
}

#endif
