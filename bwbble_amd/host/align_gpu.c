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

/* reads per GPU batch (BWB_CHUNK).  Batches are streamed as slices that park their unfinished reads for the next slice, so the
 * size no longer decides how much of the GPU idles at the end of a batch; it trades launch overhead against the balance
 * between GPUs and the memory of a slot (about 1 KB per read). */
#define GPU_CHUNK_DEFAULT (1u << 21)

static double wall(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

void set_default_aln_params(aln_params_t *p) { bwb_default_params(p); } /* align.c:22-38 */

typedef struct {
	uint64_t *aln_off; bwb_aln *alns; uint32_t n; /* host copy of one chunk's result */
	atomic_int ready;
} chunk_res_t;

typedef struct {
	int gpu, device; /* worker number; HIP device it drives */
	bwt_t *BWT;
	reads_t *reads;
	aln_params_t *params;
	uint32_t chunk;
	size_t n_chunks;
	atomic_size_t *cursor;
	chunk_res_t *res;
	const size_t *carry_src; /* per chunk: the last read before it that computes a D_seed (SIZE_MAX: none) */
	bwb_stats total;
	double kernel_ms;
} worker_t;

/* hands the finished chunk in `slot` to the writer */
static void retire(worker_t *w, bwb_hip_ctx *ctx, int slot, size_t c) {
	bwb_result r;
	if (bwb_hip_slot_result(ctx, slot, &r)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->gpu, bwb_hip_last_error());
	chunk_res_t *cr = &w->res[c];
	const uint32_t n = r.n_reads;
	cr->n = n;
	cr->aln_off = (uint64_t *)malloc(((size_t)n + 1) * 8);
	memcpy(cr->aln_off, r.aln_off, ((size_t)n + 1) * 8);
	const uint64_t tot = r.aln_off[n];
	cr->alns = (bwb_aln *)malloc((tot ? tot : 1) * sizeof(bwb_aln));
	memcpy(cr->alns, r.alns, tot * sizeof(bwb_aln));
	atomic_store(&cr->ready, 1);
}

/* One host thread per GPU.  Chunks are streamed through the slots of the context: while the search slice of chunk j runs,
 * chunk j+1 is already uploaded and queued behind it and the hits of chunk j-1 are on their way back; a slice parks its
 * unfinished reads for the next one instead of draining (include/bwbble_hip.h), so the GPU never runs a batch's tail alone. */
static void *gpu_worker(void *arg) {
	worker_t *w = (worker_t *)arg;
	bwb_hip_ctx *ctx = NULL;
	const int dbg = getenv("BWB_DEBUG") != NULL;
	double tq = wall();
	const bwtint_t hdr[5] = { w->BWT->length, w->BWT->num_words, w->BWT->num_sa, w->BWT->num_occ, w->BWT->sa0_index };
	if (bwb_hip_ctx_create(w->device, hdr, w->BWT->C, w->BWT->bwt, w->BWT->O, &ctx)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
	if (dbg) { fprintf(stderr, "[bwb host] worker %d (device %d): context + index upload %.3f s\n", w->gpu, w->device, wall() - tq); tq = wall(); }
	enum { NS = BWB_MAX_SLOTS }; /* chunks in flight: the heaviest reads of a chunk take several slices' time (they are parked and resumed), and
	                                a slot can be uploaded again only when its chunk is complete */
	size_t in_slot[NS];
	size_t j = 0; /* chunks this worker has submitted */
	for (;;) {
		const size_t c = atomic_fetch_add(w->cursor, 1);
		if (c >= w->n_chunks) break;
		const size_t r0 = c * (size_t)w->chunk;
		const uint32_t n = (uint32_t)((w->reads->count - r0) < w->chunk ? (w->reads->count - r0) : w->chunk);
		const int slot = (int)(j % NS);
		/* the last read before this chunk that computes a D_seed (longer than the seed, not dropped by -P): a short read at the
		 * head of the chunk sees its bounds, like in the reference's serial loop (inexact_match.c:35,62-65) */
		const uint8_t *carry = NULL;
		uint32_t carry_len = 0;
		if (w->carry_src[c] != SIZE_MAX) { carry = w->reads->seq + w->carry_src[c] * w->reads->stride; carry_len = w->reads->len[w->carry_src[c]]; }
		if (bwb_hip_slot_upload(ctx, slot, w->params, w->reads->seq + r0 * w->reads->stride, w->reads->len + r0, n, w->reads->stride, carry, carry_len) ||
		    bwb_hip_slot_submit(ctx, slot))
			bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
		in_slot[slot] = c;
		j++;
		if (j >= NS) retire(w, ctx, (int)((j - NS) % NS), in_slot[(j - NS) % NS]); /* NS - 1 slices stay queued while the host waits */
		if (dbg) { fprintf(stderr, "[bwb host] worker %d: chunk %zu (%u reads) submitted at +%.3f s\n", w->gpu, c, n, wall() - tq); }
	}
	for (size_t k = j >= NS - 1 ? j - (NS - 1) : 0; k < j; k++) retire(w, ctx, (int)(k % NS), in_slot[k % NS]);
	bwb_stats st;
	if (bwb_hip_flush(ctx) || bwb_hip_get_stats(ctx, &st)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
	w->total = st;
	w->kernel_ms = st.ms_calc_d + st.ms_search;
	bwb_hip_ctx_destroy(ctx);
	if (dbg) fprintf(stderr, "[bwb host] worker %d: done %.3f s\n", w->gpu, wall() - tq);
	return NULL;
}

int align_reads_inexact_gpu(bwt_t *BWT, reads_t *reads, void *precalc_sa_intervals, aln_params_t *params, char *alnFname, int n_gpus) {
	(void)precalc_sa_intervals;
	printf("BWBBLE Inexact Alignment (MI355X)...\n");
	FILE *alnFile = fopen(alnFname, "a+b");                                   /* inexact_match.c:94 */
	if (!alnFile) { perror(alnFname); bwb_die("align_reads_inexact: Cannot open ALN file: %s!", alnFname); }
	if (bwb_hip_abi_version() != BWB_HIP_ABI_VERSION) bwb_die("align_reads_inexact_gpu: libbwbble_hip.so implements C-ABI version %d, this binary was compiled against %d", bwb_hip_abi_version(), BWB_HIP_ABI_VERSION);
	const int ndev = bwb_hip_device_count();
	if (ndev < 1) bwb_die("align_reads_inexact_gpu: no HIP device found (this build has no CPU alignment path)");
	/* BWB_DEVICE_MAP=d0,d1,...: worker g drives HIP device d_g (default g).  Several workers on one device are allowed - that
	 * is how the multi-worker path (shared cursor, ordered writer) is tested on a one-GPU box. */
	int devmap[64];
	int nmap = 0;
	if (getenv("BWB_DEVICE_MAP")) {
		const char *q = getenv("BWB_DEVICE_MAP");
		while (*q && nmap < 64) {
			char *end;
			const long d = strtol(q, &end, 10);
			if (end == q) break;
			if (d < 0 || d >= ndev) bwb_die("align_reads_inexact_gpu: BWB_DEVICE_MAP names device %ld, only %d present", d, ndev);
			devmap[nmap++] = (int)d;
			q = *end == ',' ? end + 1 : end;
		}
	}
	if (n_gpus < 1) n_gpus = 1;
	if (n_gpus > (nmap ? nmap : ndev)) bwb_die("align_reads_inexact_gpu: -g %d asked for, %d HIP device(s) available", n_gpus, nmap ? nmap : ndev);
	uint32_t chunk = GPU_CHUNK_DEFAULT;
	if (getenv("BWB_CHUNK")) chunk = (uint32_t)strtoul(getenv("BWB_CHUNK"), NULL, 10);
	if (chunk < 1) chunk = 1;
	const size_t n_chunks = (reads->count + (size_t)chunk - 1) / chunk;
	chunk_res_t *res = (chunk_res_t *)calloc(n_chunks ? n_chunks : 1, sizeof(chunk_res_t));
	/* One pass over the whole file (the serial reference's look-back has no bound, inexact_match.c:35,62-65): for every chunk, the
	 * last read before it that is longer than the seed and not dropped by -P. */
	size_t *carry_src = (size_t *)malloc((n_chunks ? n_chunks : 1) * sizeof(size_t));
	{
		size_t last = SIZE_MAX;
		for (size_t q = 0; q < reads->count; q++) {
			if (q % chunk == 0) carry_src[q / chunk] = last;
			const uint32_t lq = reads->len[q];
			const uint8_t *sq = reads->seq + q * reads->stride;
			int ok = params->seed_length && lq > (uint32_t)params->seed_length && lq <= 255;
			if (ok && params->use_precalc) { if (lq < 12) ok = 0; for (int k = 0; ok && k < 12; k++) if (sq[k] > 3) ok = 0; }
			if (ok) last = q;
		}
	}
	atomic_size_t cursor = 0;
	worker_t *ws = (worker_t *)calloc((size_t)n_gpus, sizeof(worker_t));
	pthread_t *th = (pthread_t *)calloc((size_t)n_gpus, sizeof(pthread_t));
	const double t0 = wall();
	for (int g = 0; g < n_gpus; g++) {
		ws[g] = (worker_t){ .gpu = g, .device = nmap ? devmap[g] : g, .BWT = BWT, .reads = reads, .params = params, .chunk = chunk, .n_chunks = n_chunks, .cursor = &cursor, .res = res, .carry_src = carry_src };
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
	free(ws); free(th); free(res); free(carry_src);
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
	if (params->use_precalc) { /* align.c:59-65: the first `align -P` on an index leaves <fasta>.pre behind (precalc.c: written, never read) */
		char *preFname = (char *)malloc(L);
		snprintf(preFname, L, "%s.pre", fastaFname);
		FILE *pf = fopen(preFname, "r");
		if (pf) { /* load_precalc_sa_intervals (align.c:226-238) would read it: here only its shape is checked (precalc.c) */
			fclose(pf);
			if (check_precalc_file(preFname)) fprintf(stderr, "warning: %s is not a complete table of 16777216 interval lists (a run of the reference would fail on it): delete it to have it rebuilt\n", preFname);
		} else {
			t = wall();
			precalc_sa_intervals(BWT, params, preFname);
			printf("Total pre-calculated intervals time: %.2f sec\n", wall() - t);
		}
		free(preFname);
	}
	t = wall();
	align_reads_inexact_gpu(BWT, reads, NULL, params, alnsFname, n_gpus);   /* the seam: align.c:72-76 */
	printf("Total read alignment time: %.2f sec\n", wall() - t);
	free_bwt(BWT);
	free_reads(reads);
	free(bwtFname);
	return 0;
}
