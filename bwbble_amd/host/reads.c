/* reads.c - FASTQ reader with the reference's encoding (mg-aligner/io.c:410-515, tables io.h:108-130).
 * The reference mallocs three buffers per read; here the file is read once and the codes go into one
 * [count][stride] array that is handed to the GPU as is. */
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include "bwb_host.h"

/* stage clocks of the streaming reader (hostbench prints them): record scan, output allocation, base encoding */
double fq_prof_s[3];
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

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
#pragma omp parallel for schedule(static) num_threads(bwb_host_team())
	for (long i = 0; i < (long)n; i++) {
		uint8_t *d = R->seq + (size_t)i * R->stride;
		const char *s = raw + soff[i];
		for (int k = 0; k < R->len[i]; k++) d[k] = nt4(s[k]);   /* io.c:467 */
	}
	free(soff);
	printf("Loaded %d reads from %s.\n", R->count, readsFname);
	return R;
}

/* ---- the same parser as a stream of chunks (`bwbble align`): the file is mapped and taken in REGIONS; a region's record boundaries are
 * found by SEVERAL threads, each scanning one part of it (round 5; one sequential scan fed 7.7 M reads/s: enough for one GPU at -n 3, not
 * for eight at -n 0).  A FASTQ cannot be cut safely at an arbitrary byte - '@' and '+' are quality characters too - so a thread GUESSES
 * where its part's first record starts (a line that starts with '@' and whose second-next line starts with '+') and scans on from there
 * with the reference's rules (next_record_at: io.c:430-498); the parts are then stitched in file order, and a part is only accepted when
 * the sequential scanner, coming from the end of the part before, would have found that very '@' next - otherwise (a file the heuristic
 * misreads: '@' inside a line, stray text between records) the stretch is scanned again sequentially.  The records are therefore exactly
 * those of fastq2reads, whatever the file looks like.  The bases are encoded by all cores.  The reference loads the whole file before the
 * first read is aligned (io.c:410-515, align.c:55). */
struct fq_stream {
	char *raw; long sz, p; char *fname;
	size_t *soff; uint16_t *rlen; size_t n_rec, next_rec, cap; /* records found and not yet handed out: [next_rec, n_rec) */
	uint64_t count;
	int threads; long region;
};

/* next_record without the exits (a speculative scan may start in the middle of nowhere): 1 = a record, 0 = no '@' left, -1 = malformed */
static int next_record_at(const char *raw, long sz, long *pp, fq_rec_t *r, long *at) {
	long p = *pp;
	{ const char *q = p < sz ? (const char *)memchr(raw + p, '@', (size_t)(sz - p)) : NULL; p = q ? q - raw : sz; }
	if (p >= sz) { *pp = p; return 0; }
	*at = p;
	p++;
	r->ns = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; }
	if (p >= sz) return -1;
	r->nl = p - r->ns;
	if (r->nl > MAX_SEQ_NAME_LEN) r->nl = MAX_SEQ_NAME_LEN;
	p++;
	r->ss = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; }
	if (p >= sz) return -1;
	r->sl = p - r->ss;
	{ const char *q = (const char *)memchr(raw + p, '+', (size_t)(sz - p)); p = q ? q - raw : sz; }
	{ const char *q = p < sz ? (const char *)memchr(raw + p, '\n', (size_t)(sz - p)) : NULL; p = q ? q - raw : sz; }
	if (p >= sz) return -1;
	p++;
	r->qs = p;
	{ const char *q = (const char *)memchr(raw + p, '\n', (size_t)(sz - p)); p = q ? q - raw : sz; }
	if (p - r->qs != r->sl || r->sl > 65535) return -1;
	*pp = p;
	return 1;
}

/* first line start in [from, to) that looks like a record's name line: '@...' with a '+...' line two lines on; -1: none */
static long guess_record_start(const char *raw, long sz, long from, long to) {
	long q = from;
	if (q > 0 && raw[q - 1] != '\n') { const char *e = (const char *)memchr(raw + q, '\n', (size_t)(sz - q)); if (!e) return -1; q = e - raw + 1; }
	while (q < to && q < sz) {
		const char *e1 = (const char *)memchr(raw + q, '\n', (size_t)(sz - q));
		if (!e1) return -1;
		if (raw[q] == '@') {
			const char *e2 = (const char *)memchr(e1 + 1, '\n', (size_t)(raw + sz - (e1 + 1)));
			if (!e2) return -1;
			if (e2 + 1 < raw + sz && e2[1] == '+') return q;
		}
		q = e1 - raw + 1;
	}
	return -1;
}

typedef struct { size_t *soff; uint16_t *rlen; size_t n, cap; long first_at, end_p; int bad; } fq_part_t;
static void part_push(fq_part_t *pt, size_t so, uint16_t l) {
	if (pt->n == pt->cap) { pt->cap = pt->cap ? pt->cap * 2 : 4096; pt->soff = (size_t *)realloc(pt->soff, pt->cap * sizeof(size_t)); pt->rlen = (uint16_t *)realloc(pt->rlen, pt->cap * sizeof(uint16_t)); }
	pt->soff[pt->n] = so; pt->rlen[pt->n] = l; pt->n++;
}
static void stream_push(fq_stream *s, const size_t *so, const uint16_t *l, size_t n) {
	if (s->n_rec + n > s->cap) {
		if (s->next_rec) { /* drop what has been handed out */
			memmove(s->soff, s->soff + s->next_rec, (s->n_rec - s->next_rec) * sizeof(size_t));
			memmove(s->rlen, s->rlen + s->next_rec, (s->n_rec - s->next_rec) * sizeof(uint16_t));
			s->n_rec -= s->next_rec; s->next_rec = 0;
		}
		if (s->n_rec + n > s->cap) { s->cap = (s->n_rec + n) * 2; s->soff = (size_t *)realloc(s->soff, s->cap * sizeof(size_t)); s->rlen = (uint16_t *)realloc(s->rlen, s->cap * sizeof(uint16_t)); }
	}
	memcpy(s->soff + s->n_rec, so, n * sizeof(size_t)); memcpy(s->rlen + s->n_rec, l, n * sizeof(uint16_t));
	s->n_rec += n;
}

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
	long nc = sysconf(_SC_NPROCESSORS_ONLN);
	s->threads = getenv("BWB_FQ_THREADS") ? atoi(getenv("BWB_FQ_THREADS")) : (int)(nc < 1 ? 1 : (nc > 16 ? 16 : nc));
	if (s->threads < 1) s->threads = 1;
	if (s->threads > 64) s->threads = 64;
	s->region = getenv("BWB_FQ_REGION") ? atol(getenv("BWB_FQ_REGION")) : (256l << 20); /* (BWB_FQ_REGION: a test knob - regions of a few hundred bytes put a part boundary into every record) */
	if (s->region < 1) s->region = 1;
	return s;
}

/* the records of the next region of the file -> s->soff / s->rlen; 0 at the end of the file */
static int fq_scan_region(fq_stream *s) {
	const char *raw = s->raw;
	const long sz = s->sz;
	if (s->p >= sz) return 0;
	const long p0 = s->p, e = p0 + s->region < sz ? p0 + s->region : sz;
	const int T = s->threads;
	{ /* a cold file: ask for this region and the next one now (asynchronous read-ahead), instead of having sixteen scanner threads fault
	   * their parts in 128 KB at a time */
		const long pg = sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096;
		const long a0 = p0 / pg * pg, a1 = p0 + 2 * s->region < sz ? p0 + 2 * s->region : sz;
		if (a1 > a0) madvise((void *)(raw + a0), (size_t)(a1 - a0), MADV_WILLNEED);
	}
	fq_part_t *parts = (fq_part_t *)calloc((size_t)T, sizeof(fq_part_t));
	long *bnd = (long *)malloc(((size_t)T + 1) * sizeof(long));
	for (int t = 0; t <= T; t++) bnd[t] = p0 + (long)(((__int128)(e - p0) * t) / T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
	for (int t = 0; t < T; t++) {
		fq_part_t *pt = &parts[t];
		pt->first_at = -1;
		long p = t == 0 ? p0 : guess_record_start(raw, sz, bnd[t], bnd[t + 1]);
		if (p < 0) { pt->bad = 1; continue; }
		pt->end_p = p;
		for (;;) { /* the records whose '@' lies in [bnd[t], bnd[t+1]) */
			fq_rec_t rc; long at = -1, q = p;
			const int k = next_record_at(raw, sz, &q, &rc, &at);
			if (k == 0 || (k != 0 && at >= bnd[t + 1])) break;
			if (k < 0) { pt->bad = 2; break; } /* (as far as it got is kept: the stitcher takes over from end_p and reports the error) */
			if (pt->first_at < 0) pt->first_at = at;
			part_push(pt, (size_t)rc.ss, (uint16_t)rc.sl);
			p = q; pt->end_p = p;
		}
	}
	/* stitch: `cur` is where the sequential scanner stands */
	long cur = p0;
	for (int t = 0; t < T; t++) {
		fq_part_t *pt = &parts[t];
		long nat; { const char *q = cur < sz ? (const char *)memchr(raw + cur, '@', (size_t)(sz - cur)) : NULL; nat = q ? q - raw : sz; }
		if (pt->n && pt->first_at == nat && pt->bad != 1) { stream_push(s, pt->soff, pt->rlen, pt->n); cur = pt->end_p; if (pt->bad != 2) continue; }
		/* what the sequential scanner finds from `cur` up to the part's end (all of it, when the guess did not hold) */
		for (;;) {
			fq_rec_t rc; long at = -1, q = cur;
			const int k = next_record_at(raw, sz, &q, &rc, &at);
			if (k == 0 || at >= bnd[t + 1]) break;
			if (k < 0) { long pp = cur; next_record(raw, sz, &pp, &rc, s->fname); bwb_die("Error: The input file %s is not in the FASTQ format.", s->fname); } /* (next_record names the error and exits) */
			const size_t so = (size_t)rc.ss; const uint16_t l = (uint16_t)rc.sl;
			stream_push(s, &so, &l, 1);
			cur = q;
		}
	}
	if (e >= sz) { /* the last region: whatever follows the last record holds no '@' - or a broken record, which the reference reports */
		fq_rec_t rc; long at = -1, q = cur;
		const int k = next_record_at(raw, sz, &q, &rc, &at);
		if (k < 0) { long pp = cur; next_record(raw, sz, &pp, &rc, s->fname); bwb_die("Error: The input file %s is not in the FASTQ format.", s->fname); }
		if (k > 0) { const size_t so = (size_t)rc.ss; const uint16_t l = (uint16_t)rc.sl; stream_push(s, &so, &l, 1); cur = q; s->p = cur; }
		else s->p = sz;
	} else s->p = cur > p0 ? cur : p0;
	if (s->p == p0 && e < sz) { /* a region smaller than one record: take that record sequentially (progress) */
		fq_rec_t rc; long pp = p0;
		if (next_record(raw, sz, &pp, &rc, s->fname)) { const size_t so = (size_t)rc.ss; const uint16_t l = (uint16_t)rc.sl; stream_push(s, &so, &l, 1); s->p = pp; }
		else s->p = sz;
	}
	for (int t = 0; t < T; t++) { free(parts[t].soff); free(parts[t].rlen); }
	free(parts); free(bnd);
	return 1;
}

int fq_next_chunk(fq_stream *s, uint32_t max_reads, fq_chunk_t *out) {
	memset(out, 0, sizeof(*out));
	double t0 = now_s();
	while (s->n_rec - s->next_rec < max_reads && fq_scan_region(s)) { }
	fq_prof_s[0] += now_s() - t0; t0 = now_s();
	size_t avail = s->n_rec - s->next_rec;
	if (avail == 0) return 0;
	const uint32_t n = avail < max_reads ? (uint32_t)avail : max_reads;
	uint16_t *len = (uint16_t *)malloc((size_t)n * sizeof(uint16_t));
	memcpy(len, s->rlen + s->next_rec, (size_t)n * sizeof(uint16_t));
	uint32_t max_len = 0;
	for (uint32_t i = 0; i < n; i++) if (len[i] > max_len) max_len = len[i];
	const uint32_t stride = max_len ? max_len : 1;
	uint8_t *seq = (uint8_t *)malloc((size_t)n * stride);
	const char *raw = s->raw;
	const size_t *soff = s->soff + s->next_rec;
	fq_prof_s[1] += now_s() - t0; t0 = now_s();
#pragma omp parallel for schedule(static) num_threads(bwb_host_team())
	for (long i = 0; i < (long)n; i++) {
		uint8_t *d = seq + (size_t)i * stride;
		const char *q = raw + soff[i];
		int k = 0;
		for (; k < len[i]; k++) d[k] = nt4(q[k]);   /* io.c:467 */
		for (; k < (int)stride; k++) d[k] = 4;
	}
	fq_prof_s[2] += now_s() - t0;
	s->next_rec += n;
	out->n = n; out->stride = stride; out->max_len = max_len; out->seq = seq; out->len = len;
	s->count += n;
	return 1;
}

void fq_close(fq_stream *s) {
	if (!s) return;
	if (s->raw && s->sz > 0) munmap(s->raw, (size_t)s->sz);
	free(s->soff); free(s->rlen); free(s->fname); free(s);
}

void free_reads(reads_t *R) {
	if (!R) return;
	free(R->seq); free(R->len); free(R->raw); free(R->name_off); free(R->qual_off); free(R->name_len); free(R);
}
