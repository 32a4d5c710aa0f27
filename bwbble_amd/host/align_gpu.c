/*
 * align_gpu.c - `bwbble align` driver: align_reads (mg-aligner/align.c:40-87) and the GPU replacement for
 * align_reads_inexact_parallel (mg-aligner/inexact_match.c:92-168).
 *
 * Same contract as the reference pair (inexact_match.h:39-40): the caller owns BWT, reads and params; one
 * .aln record per read is appended in input order (empty records included); returns 0; errors printf + exit(1).
 * The FM-index is replicated on every GPU; reads are cut into contiguous chunks that the per-GPU host threads
 * pull from a shared cursor (the per-read work is heavy-tailed) and a writer emits them in order.
 * No collective is involved.  There is NO CPU alignment path: without a GPU this exits with an error.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "bwb_host.h"

/* reads per GPU batch (BWB_CHUNK): per-read work is heavy-tailed, so a batch ends with a drain phase in which few lanes
 * are busy; measured at chr21 scale, -n 3: 0.75 M reads/s with 1 M-read batches, 1.19 M reads/s with 4 M-read ones */
#define GPU_CHUNK_DEFAULT (1u << 22)

static double wall(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

void set_default_aln_params(aln_params_t *p) { bwb_default_params(p); } /* align.c:22-38 */

typedef struct {
	uint64_t *aln_off; bwb_aln *alns; uint32_t n; /* host copy of one chunk's result */
	atomic_int ready;
} chunk_res_t;

typedef struct {
	int gpu;
	bwt_t *BWT;
	reads_t *reads;
	aln_params_t *params;
	uint32_t chunk;
	size_t n_chunks;
	atomic_size_t *cursor;
	chunk_res_t *res;
	bwb_stats total;
	double kernel_ms;
} worker_t;

static void *gpu_worker(void *arg) {
	worker_t *w = (worker_t *)arg;
	bwb_hip_ctx *ctx = NULL;
	const int dbg = getenv("BWB_DEBUG") != NULL;
	double tq = wall();
	const bwtint_t hdr[5] = { w->BWT->length, w->BWT->num_words, w->BWT->num_sa, w->BWT->num_occ, w->BWT->sa0_index };
	if (bwb_hip_ctx_create(w->gpu, hdr, w->BWT->C, w->BWT->bwt, w->BWT->O, &ctx)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->gpu, bwb_hip_last_error());
	if (dbg) { fprintf(stderr, "[bwb host] GPU %d: context + index upload %.3f s\n", w->gpu, wall() - tq); tq = wall(); }
	for (;;) {
		const size_t c = atomic_fetch_add(w->cursor, 1);
		if (c >= w->n_chunks) break;
		const size_t r0 = c * (size_t)w->chunk;
		const uint32_t n = (uint32_t)((w->reads->count - r0) < w->chunk ? (w->reads->count - r0) : w->chunk);
		bwb_result r;
		if (bwb_hip_align_batch(ctx, w->params, w->reads->seq + r0 * w->reads->stride, w->reads->len + r0, n, w->reads->stride, &r))
			bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->gpu, bwb_hip_last_error());
		bwb_stats st;
		bwb_hip_get_stats(ctx, &st);
		w->total.visits_single += st.visits_single; w->total.visits_alphabet += st.visits_alphabet;
		w->total.heap_pops += st.heap_pops; w->total.heap_pushes += st.heap_pushes; w->total.n_alignments += st.n_alignments;
		w->total.n_overflow_reads += st.n_overflow_reads;
		w->kernel_ms += st.ms_calc_d + st.ms_search;
		chunk_res_t *cr = &w->res[c];
		cr->n = n;
		cr->aln_off = (uint64_t *)malloc(((size_t)n + 1) * 8);
		memcpy(cr->aln_off, r.aln_off, ((size_t)n + 1) * 8);
		const uint64_t tot = r.aln_off[n];
		cr->alns = (bwb_aln *)malloc((tot ? tot : 1) * sizeof(bwb_aln));
		memcpy(cr->alns, r.alns, tot * sizeof(bwb_aln));
		atomic_store(&cr->ready, 1);
		if (dbg) { fprintf(stderr, "[bwb host] GPU %d: chunk %zu (%u reads) %.3f s\n", w->gpu, c, n, wall() - tq); tq = wall(); }
	}
	bwb_hip_ctx_destroy(ctx);
	if (dbg) fprintf(stderr, "[bwb host] GPU %d: context destroy %.3f s\n", w->gpu, wall() - tq);
	return NULL;
}

int align_reads_inexact_gpu(bwt_t *BWT, reads_t *reads, void *precalc_sa_intervals, aln_params_t *params, char *alnFname, int n_gpus) {
	(void)precalc_sa_intervals;
	printf("BWBBLE Inexact Alignment (MI355X)...\n");
	FILE *alnFile = fopen(alnFname, "a+b");                                   /* inexact_match.c:94 */
	if (!alnFile) { perror(alnFname); bwb_die("align_reads_inexact: Cannot open ALN file: %s!", alnFname); }
	const int ndev = bwb_hip_device_count();
	if (ndev < 1) bwb_die("align_reads_inexact_gpu: no HIP device found (this build has no CPU alignment path)");
	if (n_gpus < 1 || n_gpus > ndev) n_gpus = n_gpus < 1 ? 1 : ndev;
	uint32_t chunk = GPU_CHUNK_DEFAULT;
	if (getenv("BWB_CHUNK")) chunk = (uint32_t)strtoul(getenv("BWB_CHUNK"), NULL, 10);
	if (chunk < 1) chunk = 1;
	const size_t n_chunks = (reads->count + (size_t)chunk - 1) / chunk;
	chunk_res_t *res = (chunk_res_t *)calloc(n_chunks ? n_chunks : 1, sizeof(chunk_res_t));
	atomic_size_t cursor = 0;
	worker_t *ws = (worker_t *)calloc((size_t)n_gpus, sizeof(worker_t));
	pthread_t *th = (pthread_t *)calloc((size_t)n_gpus, sizeof(pthread_t));
	const double t0 = wall();
	for (int g = 0; g < n_gpus; g++) {
		ws[g] = (worker_t){ .gpu = g, .BWT = BWT, .reads = reads, .params = params, .chunk = chunk, .n_chunks = n_chunks, .cursor = &cursor, .res = res };
		if (pthread_create(&th[g], NULL, gpu_worker, &ws[g])) bwb_die("align_reads_inexact_gpu: cannot start a host thread");
	}
	/* ordered writer (the reference writes after each batch, inexact_match.c:154-162) */
	size_t processed = 0;
	for (size_t c = 0; c < n_chunks; c++) {
		while (!atomic_load(&res[c].ready)) usleep(200);
		for (uint32_t i = 0; i < res[c].n; i++)
			alns2alnf_bin(res[c].alns + res[c].aln_off[i], res[c].aln_off[i + 1] - res[c].aln_off[i], alnFile);
		processed += res[c].n;
		printf("Processed %zu reads. Elapsed: %.2f sec\n", processed, wall() - t0);
		free(res[c].aln_off); free(res[c].alns);
	}
	bwb_stats tot; memset(&tot, 0, sizeof(tot));
	double kms = 0;
	for (int g = 0; g < n_gpus; g++) {
		pthread_join(th[g], NULL);
		tot.visits_single += ws[g].total.visits_single; tot.visits_alphabet += ws[g].total.visits_alphabet;
		tot.heap_pops += ws[g].total.heap_pops; tot.n_alignments += ws[g].total.n_alignments; tot.n_overflow_reads += ws[g].total.n_overflow_reads;
		if (ws[g].kernel_ms > kms) kms = ws[g].kernel_ms;
	}
	const double dt = wall() - t0;
	printf("GPUs: %d  reads: %u  wall: %.3f sec (%.0f reads/s incl. index upload)  kernel: %.1f ms  rank-block visits: %llu  hits: %llu  re-run reads: %llu\n",
	       n_gpus, reads->count, dt, reads->count / (dt > 0 ? dt : 1), kms, (unsigned long long)(tot.visits_single + tot.visits_alphabet),
	       (unsigned long long)tot.n_alignments, (unsigned long long)tot.n_overflow_reads);
	free(ws); free(th); free(res);
	fclose(alnFile);
	return 0;
}

int align_reads(char *fastaFname, char *readsFname, char *alnsFname, aln_params_t *params, int n_gpus) { /* align.c:40-87 */
	printf("**** BWBBLE Read Alignment ****\n");
	size_t L = strlen(fastaFname) + 8;
	char *bwtFname = (char *)malloc(L);
	snprintf(bwtFname, L, "%s.bwt", fastaFname);
	remove(alnsFname); /* align.c:48 */
	double t = wall();
	bwt_t *BWT = load_bwt(bwtFname, 0);
	printf("Total BWT loading time: %.2f sec\n", wall() - t);
	t = wall();
	reads_t *reads = fastq2reads(readsFname);
	printf("Total read loading time: %.2f sec\n", wall() - t);
	t = wall();
	align_reads_inexact_gpu(BWT, reads, NULL, params, alnsFname, n_gpus);   /* the seam: align.c:72-76 */
	printf("Total read alignment time: %.2f sec\n", wall() - t);
	free_bwt(BWT);
	free_reads(reads);
	free(bwtFname);
	return 0;
}
