/* reads.c - FASTQ reader with the reference's encoding (mg-aligner/io.c:410-515, tables io.h:108-130).
 * The reference mallocs three buffers per read; here the file is read once and the codes go into one
 * [count][stride] array that is handed to the GPU as is. */
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "bwb_host.h"

static inline uint8_t nt4(char c) { /* nt4_table io.h:113-130: A0 G1 C2 T3, everything else 4 */
	switch (c) {
	case 'A': case 'a': return 0;
	case 'G': case 'g': return 1;
	case 'C': case 'c': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

/* one FASTQ record, located with the reference's scanning rules (io.c:430-498): skip to the next '@', the name line, the sequence
 * line, skip to the next '+', the rest of that line, the quality line (as long as the sequence).  Offsets into raw. */
typedef struct { long ns, nl, ss, sl, qs; } fq_rec_t;
static int next_record(const char *raw, long sz, long *pp, fq_rec_t *r, const char *fname) {
	long p = *pp;
	while (p < sz && raw[p] != '@') p++;                 /* io.c:430-434 */
	if (p >= sz) { *pp = p; return 0; }
	p++;
	r->ns = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; } /* line 1 */
	if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", fname);
	r->nl = p - r->ns;
	if (r->nl > MAX_SEQ_NAME_LEN) r->nl = MAX_SEQ_NAME_LEN;    /* io.c:439 */
	p++;
	r->ss = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; } /* line 2 */
	if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", fname);
	r->sl = p - r->ss;
	{ const char *q = (const char *)memchr(raw + p, '+', (size_t)(sz - p)); p = q ? q - raw : sz; }    /* io.c:474-477 */
	{ const char *q = p < sz ? (const char *)memchr(raw + p, '\n', (size_t)(sz - p)) : NULL; p = q ? q - raw : sz; } /* line 3 */
	if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", fname);
	p++;
	r->qs = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; } /* line 4 */
	if (p - r->qs != r->sl) bwb_die("Error: The number of quality score symbols does not match the length of the read sequence."); /* io.c:495-498 */
	if (r->sl > 65535) bwb_die("Error: read longer than 65535 bases.");
	*pp = p;
	return 1;
}

reads_t *fastq2reads(const char *readsFname) {
	FILE *f = fopen(readsFname, "r");
	if (!f) bwb_die("load_reads_fastq: Cannot open reads file: %s !", readsFname);
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	reads_t *R = (reads_t *)calloc(1, sizeof(reads_t));
	R->raw = (char *)malloc((size_t)sz + 1);
	if (!R->raw || fread(R->raw, 1, (size_t)sz, f) != (size_t)sz) bwb_die("load_reads_fastq: Cannot read reads file: %s !", readsFname);
	fclose(f);
	R->raw[sz] = 0;
	const char *raw = R->raw;
	size_t cap = 1u << 16, n = 0;
	size_t *soff = (size_t *)malloc(cap * sizeof(size_t));
	R->name_off = (size_t *)malloc(cap * sizeof(size_t));
	R->qual_off = (size_t *)malloc(cap * sizeof(size_t));
	R->name_len = (uint16_t *)malloc(cap * sizeof(uint16_t));
	R->len = (uint16_t *)malloc(cap * sizeof(uint16_t));
	long p = 0;
	fq_rec_t rc;
	while (next_record(raw, sz, &p, &rc, readsFname)) {
		if (n == cap) {
			cap *= 2;
			soff = (size_t *)realloc(soff, cap * sizeof(size_t));
			R->name_off = (size_t *)realloc(R->name_off, cap * sizeof(size_t));
			R->qual_off = (size_t *)realloc(R->qual_off, cap * sizeof(size_t));
			R->name_len = (uint16_t *)realloc(R->name_len, cap * sizeof(uint16_t));
			R->len = (uint16_t *)realloc(R->len, cap * sizeof(uint16_t));
		}
		soff[n] = (size_t)rc.ss; R->name_off[n] = (size_t)rc.ns; R->name_len[n] = (uint16_t)rc.nl; R->qual_off[n] = (size_t)rc.qs;
		R->len[n] = (uint16_t)rc.sl;
		if ((unsigned)rc.sl > R->max_len) R->max_len = (unsigned)rc.sl;
		n++;
	}
	R->count = (unsigned)n;
	R->stride = R->max_len ? R->max_len : 1;
	R->seq = (uint8_t *)malloc((n ? n : 1) * (size_t)R->stride);
	memset(R->seq, 4, (n ? n : 1) * (size_t)R->stride);
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)n; i++) {
		uint8_t *d = R->seq + (size_t)i * R->stride;
		const char *s = raw + soff[i];
		for (int k = 0; k < R->len[i]; k++) d[k] = nt4(s[k]);   /* io.c:467 */
	}
	free(soff);
	printf("Loaded %d reads from %s.\n", R->count, readsFname);
	return R;
}

/* ---- the same parser as a stream of chunks (`bwbble align`): the file is mapped, the record boundaries of a chunk are found by the
 * sequential scan above (a FASTQ cannot be cut safely anywhere else: '@' and '+' are quality characters too), the bases are encoded
 * by all cores.  The reference loads the whole file before the first read is aligned (io.c:410-515, align.c:55). */
struct fq_stream { char *raw; long sz, p; char *fname; size_t *soff; size_t cap; uint64_t count; };

fq_stream *fq_open(const char *readsFname) {
	const int fd = open(readsFname, O_RDONLY);
	if (fd < 0) bwb_die("load_reads_fastq: Cannot open reads file: %s !", readsFname);
	struct stat st;
	if (fstat(fd, &st)) bwb_die("load_reads_fastq: Cannot read reads file: %s !", readsFname);
	fq_stream *s = (fq_stream *)calloc(1, sizeof(fq_stream));
	s->sz = (long)st.st_size; s->fname = strdup(readsFname);
	if (s->sz > 0) {
		s->raw = (char *)mmap(NULL, (size_t)s->sz, PROT_READ, MAP_PRIVATE, fd, 0);
		if (s->raw == MAP_FAILED) bwb_die("load_reads_fastq: Cannot read reads file: %s !", readsFname);
		madvise(s->raw, (size_t)s->sz, MADV_SEQUENTIAL);
	}
	close(fd);
	return s;
}

int fq_next_chunk(fq_stream *s, uint32_t max_reads, fq_chunk_t *out) {
	memset(out, 0, sizeof(*out));
	if (s->cap < max_reads) { s->cap = max_reads; s->soff = (size_t *)realloc(s->soff, s->cap * sizeof(size_t)); }
	uint16_t *len = (uint16_t *)malloc((size_t)(max_reads ? max_reads : 1) * sizeof(uint16_t));
	uint32_t n = 0, max_len = 0;
	fq_rec_t rc;
	while (n < max_reads && next_record(s->raw, s->sz, &s->p, &rc, s->fname)) {
		s->soff[n] = (size_t)rc.ss; len[n] = (uint16_t)rc.sl;
		if ((uint32_t)rc.sl > max_len) max_len = (uint32_t)rc.sl;
		n++;
	}
	if (n == 0) { free(len); return 0; }
	const uint32_t stride = max_len ? max_len : 1;
	uint8_t *seq = (uint8_t *)malloc((size_t)n * stride);
	const char *raw = s->raw;
	const size_t *soff = s->soff;
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)n; i++) {
		uint8_t *d = seq + (size_t)i * stride;
		const char *q = raw + soff[i];
		int k = 0;
		for (; k < len[i]; k++) d[k] = nt4(q[k]);   /* io.c:467 */
		for (; k < (int)stride; k++) d[k] = 4;
	}
	out->n = n; out->stride = stride; out->max_len = max_len; out->seq = seq; out->len = len;
	s->count += n;
	return 1;
}

void fq_close(fq_stream *s) {
	if (!s) return;
	if (s->raw && s->sz > 0) munmap(s->raw, (size_t)s->sz);
	free(s->soff); free(s->fname); free(s);
}

void free_reads(reads_t *R) {
	if (!R) return;
	free(R->seq); free(R->len); free(R->raw); free(R->name_off); free(R->qual_off); free(R->name_len); free(R);
}
