/* bwt_io.c - .bwt file I/O, byte-compatible with the reference (mg-aligner/bwt.c:66-125):
 * u64 length, num_words, num_sa, num_occ, sa0_index, u64 C[17], u32 bwt[num_words], u64 O[num_occ*16], u64 SA[num_sa]. */
#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "bwb_host.h"

void bwb_die(const char *fmt, ...) { /* the reference printf()s and exit(1)s on every error (e.g. bwt.c:68-70) */
	va_list ap;
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
	printf("\n");
	exit(1);
}

void store_bwt(const bwt_t *BWT, const char *bwtFname) {
	FILE *f = fopen(bwtFname, "wb");
	if (!f) bwb_die("store_bwt: Cannot open the BWT file %s!", bwtFname);
	const bwtint_t hdr[5] = { BWT->length, BWT->num_words, BWT->num_sa, BWT->num_occ, BWT->sa0_index };
	fwrite(hdr, sizeof(bwtint_t), 5, f);
	fwrite(BWT->C, sizeof(bwtint_t), ALPHABET_SIZE + 1, f);
	fwrite(BWT->bwt, sizeof(uint32_t), BWT->num_words, f);
	fwrite(BWT->O, sizeof(bwtint_t), BWT->num_occ * ALPHABET_SIZE, f);
	fwrite(BWT->SA, sizeof(bwtint_t), BWT->num_sa, f);
	fclose(f);
}

/* ---- loading: the file is read by several threads in units of LOAD_UNIT blocks, in file order; blocks_ready follows the leading
 * complete units, so that a GPU context (bwb_hip_ctx_create_streamed) can upload the head of the index while the tail is still being
 * read.  The reference freads the file from one thread (bwt.c:90-125): 12 GB at GRCh37 scale. */
#define LOAD_UNIT_DEFAULT (1ull << 17) /* 128-character blocks per unit: 8 MB of bwt words + 16 MB of O rows */
static uint64_t load_unit(void) { /* (BWB_LOAD_UNIT: a test knob - a unit of a few blocks runs the multi-unit logic on the toy index) */
	const char *e = getenv("BWB_LOAD_UNIT");
	const uint64_t v = e ? strtoull(e, NULL, 10) : 0;
	return v ? v : LOAD_UNIT_DEFAULT;
}
typedef struct {
	bwt_t *B;
	int fd;
	char *fname;
	uint64_t n_units, next_unit, off_bwt, off_O, off_SA, unit;
	int load_sa, sa_done;
	unsigned char *done;
	pthread_mutex_t mu;
	pthread_t th[16];
	int n_threads;
	double t0;
} bwt_loader_t;

static void pread_all(int fd, void *buf, size_t n, uint64_t off, const char *fname) {
	char *p = (char *)buf;
	while (n) {
		const ssize_t k = pread(fd, p, n, (off_t)off);
		if (k <= 0) bwb_die("load_bwt: Could not read BWT from file: %s!", fname);
		p += k; off += (uint64_t)k; n -= (size_t)k;
	}
}

static void *bwt_loader_thread(void *arg) {
	bwt_loader_t *L = (bwt_loader_t *)arg;
	bwt_t *B = L->B;
	for (;;) {
		pthread_mutex_lock(&L->mu);
		const uint64_t u = L->next_unit++;
		int do_sa = 0;
		if (u >= L->n_units && L->load_sa && !L->sa_done) { L->sa_done = 1; do_sa = 1; }
		pthread_mutex_unlock(&L->mu);
		if (u >= L->n_units) {
			if (do_sa) pread_all(L->fd, B->SA, B->num_sa * sizeof(bwtint_t), L->off_SA, L->fname);
			{ /* (the last thread to get here names the moment the file was in memory) */
				struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
				const double now = ts.tv_sec + 1e-9 * ts.tv_nsec - L->t0;
				pthread_mutex_lock(&L->mu);
				if (now > B->load_seconds) B->load_seconds = now;
				pthread_mutex_unlock(&L->mu);
			}
			return NULL;
		}
		const uint64_t LOAD_UNIT = L->unit;
		const uint64_t b0 = u * LOAD_UNIT, nb = (B->num_occ - b0) < LOAD_UNIT ? (B->num_occ - b0) : LOAD_UNIT;
		const uint64_t w0 = b0 * 16, nw = (B->num_words - w0) < nb * 16 ? (B->num_words - w0) : nb * 16;
		pread_all(L->fd, B->bwt + w0, nw * sizeof(uint32_t), L->off_bwt + w0 * sizeof(uint32_t), L->fname);
		pread_all(L->fd, B->O + b0 * ALPHABET_SIZE, nb * ALPHABET_SIZE * sizeof(bwtint_t), L->off_O + b0 * ALPHABET_SIZE * sizeof(bwtint_t), L->fname);
		pthread_mutex_lock(&L->mu);
		L->done[u] = 1;
		uint64_t lead = B->blocks_ready / LOAD_UNIT;
		while (lead < L->n_units && L->done[lead]) lead++;
		const uint64_t ready = lead * LOAD_UNIT < B->num_occ ? lead * LOAD_UNIT : B->num_occ;
		__atomic_store_n(&B->blocks_ready, ready, __ATOMIC_RELEASE);
		pthread_mutex_unlock(&L->mu);
	}
}

bwt_t *load_bwt_start(const char *bwtFname, int loadSA) {
	const int fd = open(bwtFname, O_RDONLY);
	if (fd < 0) bwb_die("load_bwt: Cannot open the BWT file: %s!", bwtFname);
	bwt_t *B = (bwt_t *)calloc(1, sizeof(bwt_t));
	bwtint_t hdr[5 + ALPHABET_SIZE + 1];
	if (pread(fd, hdr, sizeof(hdr), 0) != (ssize_t)sizeof(hdr)) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	B->length = hdr[0]; B->num_words = hdr[1]; B->num_sa = hdr[2]; B->num_occ = hdr[3]; B->sa0_index = hdr[4];
	memcpy(B->C, hdr + 5, sizeof(bwtint_t) * (ALPHABET_SIZE + 1));
	/* the header must be consistent BEFORE anything is sized from it - the allocations below, and the read sizes a loader thread computes
	 * (bwt.c:161-218 writes exactly these; a crafted num_words < 16 * (num_occ - 1) would make a unit's word count underflow) - and the
	 * file must hold what it promises (the reference's fread checks the same, bwt.c:104-118): a crafted header cannot ask for more memory
	 * than the file has bytes */
	if (B->length < 2 || B->num_occ != (B->length + 127) / 128 || B->num_words != (B->length + 7) / 8 || B->num_sa != (B->length + 31) / 32 || B->sa0_index >= B->length)
		bwb_die("load_bwt: %s: inconsistent header (length %llu, num_words %llu, num_sa %llu, num_occ %llu)", bwtFname, (unsigned long long)B->length,
		        (unsigned long long)B->num_words, (unsigned long long)B->num_sa, (unsigned long long)B->num_occ);
	{
		struct stat st;
		const uint64_t need = sizeof(hdr) + B->num_words * sizeof(uint32_t) + B->num_occ * ALPHABET_SIZE * sizeof(bwtint_t) + (loadSA ? B->num_sa * sizeof(bwtint_t) : 0);
		if (fstat(fd, &st) || (uint64_t)st.st_size < need) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	}
	B->bwt = (uint32_t *)malloc((B->num_words ? B->num_words : 1) * sizeof(uint32_t));
	B->O = (bwtint_t *)malloc((B->num_occ ? B->num_occ : 1) * ALPHABET_SIZE * sizeof(bwtint_t));
	if (!B->bwt || !B->O) bwb_die("load_bwt: Could not allocate memory for the BWT index. ");
	if (loadSA) {
		B->SA = (bwtint_t *)malloc((B->num_sa ? B->num_sa : 1) * sizeof(bwtint_t));
		if (!B->SA) bwb_die("load_bwt: Could not allocate memory for the BWT index. ");
	}
	bwt_loader_t *L = (bwt_loader_t *)calloc(1, sizeof(bwt_loader_t));
	L->B = B; L->fd = fd; L->fname = strdup(bwtFname); L->load_sa = loadSA;
	L->off_bwt = sizeof(hdr); L->off_O = L->off_bwt + B->num_words * sizeof(uint32_t); L->off_SA = L->off_O + B->num_occ * ALPHABET_SIZE * sizeof(bwtint_t);
	L->unit = load_unit();
	L->n_units = (B->num_occ + L->unit - 1) / L->unit;
	L->done = (unsigned char *)calloc(L->n_units ? L->n_units : 1, 1);
	pthread_mutex_init(&L->mu, NULL);
	long nc = sysconf(_SC_NPROCESSORS_ONLN);
	L->n_threads = getenv("BWB_LOAD_THREADS") ? atoi(getenv("BWB_LOAD_THREADS")) : (int)(nc < 1 ? 1 : (nc > 8 ? 8 : nc));
	if (L->n_threads < 1) L->n_threads = 1;
	if (L->n_threads > 16) L->n_threads = 16;
	B->loader = L;
	{ struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); L->t0 = ts.tv_sec + 1e-9 * ts.tv_nsec; }
	for (int t = 0; t < L->n_threads; t++)
		if (pthread_create(&L->th[t], NULL, bwt_loader_thread, L)) bwb_die("load_bwt: cannot start a loader thread");
	return B;
}

void load_bwt_wait(bwt_t *B) {
	bwt_loader_t *L = (bwt_loader_t *)B->loader;
	if (!L) return;
	for (int t = 0; t < L->n_threads; t++) pthread_join(L->th[t], NULL);
	close(L->fd);
	pthread_mutex_destroy(&L->mu);
	free(L->done); free(L->fname); free(L);
	B->loader = NULL;
	__atomic_store_n(&B->blocks_ready, B->num_occ, __ATOMIC_RELEASE);
}

double bwt_load_seconds(const bwt_t *B) { return B ? B->load_seconds : 0.0; } /* (complete once load_bwt_wait has returned or blocks_ready == num_occ) */

bwt_t *load_bwt(const char *bwtFname, int loadSA) {
	bwt_t *B = load_bwt_start(bwtFname, loadSA);
	load_bwt_wait(B);
	return B;
}

void free_bwt(bwt_t *B) {
	if (!B) return;
	load_bwt_wait(B);
	free(B->bwt); free(B->O); free(B->SA); free(B);
}
