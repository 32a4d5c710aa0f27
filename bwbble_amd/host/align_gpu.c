/*
 * align_gpu.c - `bwbble align` driver: align_reads (mg-aligner/align.c:40-87) and the GPU replacement for
 * align_reads_inexact_parallel (mg-aligner/inexact_match.c:92-168).
 *
 * Same contract as the reference pair (inexact_match.h:39-40): the caller owns BWT, reads and params; one
 * .aln record per read is appended in input order (empty records included); returns 0; errors printf + exit(1).
 * The FM-index is replicated on every GPU; reads are cut into contiguous chunks that the per-GPU host threads
 * pull from a shared queue (the per-read work is heavy-tailed) and a writer emits them in order.
 * No collective is involved.  There is NO CPU alignment path: without a GPU this exits with an error.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
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

/* One chunk of the FASTQ on its way through the pipeline: parsed by the reader thread (several scanner threads: reads.c), aligned by one
 * of the GPU workers - which also turns the hits into the chunk's .aln bytes (aln_io.c: alns2alnf_buf) -, written by the main thread in
 * file order with one write per chunk. */
typedef struct chunk {
	size_t idx;
	fq_chunk_t fq;                 /* the reads (freed once the chunk is uploaded) */
	uint8_t *carry; uint32_t carry_len; /* the last read before the chunk that computes a D_seed (NULL: none), a copy */
	unsigned char *buf; size_t buf_len; uint32_t n; /* the chunk's .aln records, serialised by the worker that received its hits */
	int ready;
	struct chunk *next;            /* production order */
} chunk_t;

typedef struct {
	pthread_mutex_t mu;
	pthread_cond_t cv_work, cv_done, cv_space;
	chunk_t *first, *last;         /* every chunk not yet written, in order */
	chunk_t *unclaimed;            /* the first chunk no worker has taken yet */
	size_t n_parsed, n_written, max_ahead;
	int reader_done;
	uint64_t n_reads;
	/* reader */
	const char *readsFname;
	const aln_params_t *params;
	uint32_t chunk_reads;
	double t_first_chunk;
} pipe_t;

typedef struct {
	int gpu, device; /* worker number; HIP device it drives */
	bwt_t *BWT;
	aln_params_t *params;
	pipe_t *pp;
	bwb_stats total;
	double kernel_ms, t_ctx, t_pool, t_first_submit; /* index upload; the chunk pool's hipMalloc; worker start to the first slice queued */
	unsigned long long pool_bytes;
	unsigned long long n_reads; /* reads of the chunks this worker took */
} worker_t;

/* The reader: chunk k+1 is parsed (record boundaries by one sequential scan, bases encoded by all cores: reads.c) while chunk k is on
 * the GPUs.  It also names, for every chunk, the last read before it that computes a D_seed - longer than the seed, not dropped by
 * -P - whose bounds a short read at the head of the chunk sees, like in the reference's serial loop (inexact_match.c:35,62-65; the
 * look-back has no bound: the read is carried along as a copy). */
static void *reader_thread(void *arg) {
	pipe_t *pp = (pipe_t *)arg;
	fq_stream *fs = fq_open(pp->readsFname);
	uint8_t *last_src = NULL; uint32_t last_len = 0;
	const aln_params_t *params = pp->params;
	const double t0 = wall();
	for (;;) {
		pthread_mutex_lock(&pp->mu);
		while (pp->n_parsed - pp->n_written >= pp->max_ahead) pthread_cond_wait(&pp->cv_space, &pp->mu);
		pthread_mutex_unlock(&pp->mu);
		chunk_t *c = (chunk_t *)calloc(1, sizeof(chunk_t));
		if (!fq_next_chunk(fs, pp->chunk_reads, &c->fq)) { free(c); break; }
		if (last_src) { c->carry = (uint8_t *)malloc(last_len); memcpy(c->carry, last_src, last_len); c->carry_len = last_len; }
		for (uint32_t q = c->fq.n; q-- > 0;) { /* the chunk's last D_seed source, if it has one */
			const uint32_t lq = c->fq.len[q];
			const uint8_t *sq = c->fq.seq + (size_t)q * c->fq.stride;
			int ok = params->seed_length && lq > (uint32_t)params->seed_length && lq <= 255;
			if (ok && params->use_precalc) { if (lq < 12) ok = 0; for (int k = 0; ok && k < 12; k++) if (sq[k] > 3) ok = 0; }
			if (ok) { last_src = (uint8_t *)realloc(last_src, lq); memcpy(last_src, sq, lq); last_len = lq; break; }
		}
		pthread_mutex_lock(&pp->mu);
		c->idx = pp->n_parsed++;
		pp->n_reads += c->fq.n;
		if (c->idx == 0) pp->t_first_chunk = wall() - t0;
		if (pp->last) pp->last->next = c; else pp->first = c;
		pp->last = c;
		if (!pp->unclaimed) pp->unclaimed = c;
		pthread_cond_broadcast(&pp->cv_work);
		pthread_mutex_unlock(&pp->mu);
	}
	pthread_mutex_lock(&pp->mu);
	pp->reader_done = 1;
	pthread_cond_broadcast(&pp->cv_work);
	pthread_cond_broadcast(&pp->cv_done);
	pthread_mutex_unlock(&pp->mu);
	free(last_src);
	fq_close(fs);
	return NULL;
}

/* hands the finished chunk in `slot` to the writer */
static void retire(worker_t *w, bwb_hip_ctx *ctx, int slot, chunk_t *c) {
	bwb_result r;
	if (bwb_hip_slot_result(ctx, slot, &r)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->gpu, bwb_hip_last_error());
	c->n = r.n_reads;
	c->buf = alns2alnf_buf(r.alns, r.aln_off, r.n_reads, &c->buf_len); /* (with all cores; the library's result buffers stay valid until the slot is uploaded again) */
	pthread_mutex_lock(&w->pp->mu);
	c->ready = 1;
	pthread_cond_broadcast(&w->pp->cv_done);
	pthread_mutex_unlock(&w->pp->mu);
}

/* the CPUs of the GPU's NUMA node, when the machine has several nodes: the worker thread, its pinned staging buffers (first touched
 * by it) and the copies into them stay next to the GPU's PCIe root */
static void pin_to_device_node(int device, int dbg) {
	const int node = bwb_hip_device_numa_node(device);
	if (node < 0) return;
	char path[96], buf[4096];
	snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
	FILE *f = fopen(path, "r");
	if (!f) return;
	if (!fgets(buf, sizeof(buf), f)) { fclose(f); return; }
	fclose(f);
	cpu_set_t set; CPU_ZERO(&set);
	int any = 0;
	for (char *q = buf; *q;) {
		char *end;
		long a = strtol(q, &end, 10), b = a;
		if (end == q) break;
		if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); }
		for (long k = a; k <= b && k < CPU_SETSIZE; k++) { CPU_SET((int)k, &set); any = 1; }
		q = *end == ',' ? end + 1 : end;
		if (*q == '\n') break;
	}
	if (any && pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0 && dbg) fprintf(stderr, "[bwb host] device %d: worker thread on NUMA node %d (cpus %s", device, node, buf);
}

/* One host thread per GPU.  Chunks are streamed through the slots of the context: while the search slice of chunk j runs,
 * chunk j+1 is already uploaded and queued behind it and the hits of chunk j-1 are on their way back; a slice parks its
 * unfinished reads for the next one instead of draining (include/bwbble_hip.h), so the GPU never runs a batch's tail alone. */
static void *gpu_worker(void *arg) {
	worker_t *w = (worker_t *)arg;
	pipe_t *pp = w->pp;
	bwb_hip_ctx *ctx = NULL;
	const int dbg = getenv("BWB_DEBUG") != NULL;
	pin_to_device_node(w->device, dbg);
	double tq = wall();
	const bwtint_t hdr[5] = { w->BWT->length, w->BWT->num_words, w->BWT->num_sa, w->BWT->num_occ, w->BWT->sa0_index };
	/* The index goes to the GPU - on a thread of the library - while the loader threads are still reading the tail of the .bwt file
	 * (bwt_io.c), and while this thread already uploads its first chunk: that first slot_upload sizes and allocates the context's scratch
	 * and its heap chunk pool (seconds of hipMalloc at GRCh37 scale, which rounds 3-5 paid AFTER the index upload); slot_submit waits for
	 * the index by itself. */
	const int sync_create = getenv("BWB_SYNC_CREATE") != NULL; /* (A/B: rounds 3-5 - the index first, everything else behind it) */
	if ((sync_create ? bwb_hip_ctx_create_streamed : bwb_hip_ctx_create_async)(w->device, hdr, w->BWT->C, w->BWT->bwt, w->BWT->O, w->BWT->loader ? &w->BWT->blocks_ready : NULL, &ctx))
		bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
	if (dbg) fprintf(stderr, "[bwb host] worker %d: context created at +%.3f s (%s)\n", w->gpu, wall() - tq, sync_create ? "index uploaded" : "index upload under way");
	enum { NS = BWB_MAX_SLOTS }; /* chunks in flight: the heaviest reads of a chunk take several slices' time (they are parked and resumed), and
	                                a slot can be uploaded again only when its chunk is complete */
	chunk_t *in_slot[NS];
	size_t j = 0, retired = 0; /* chunks this worker has submitted / handed to the writer */
	for (;;) {
		pthread_mutex_lock(&pp->mu);
		/* nothing to take: with chunks of its own in flight the worker hands its oldest one over instead of waiting (the reader may
		 * be held back by exactly that chunk: it parses at most max_ahead chunks beyond the one being written) */
		while (!pp->unclaimed && !pp->reader_done && retired == j) pthread_cond_wait(&pp->cv_work, &pp->mu);
		chunk_t *c = pp->unclaimed;
		if (c) pp->unclaimed = c->next;
		const int done = !c && pp->reader_done;
		pthread_mutex_unlock(&pp->mu);
		if (!c) {
			if (retired < j) { retire(w, ctx, (int)(retired % NS), in_slot[retired % NS]); retired++; continue; }
			if (done) break;
			continue;
		}
		const int slot = (int)(j % NS);
		if (j >= NS && retired + NS <= j) { retire(w, ctx, slot, in_slot[slot]); retired++; } /* the slot's previous chunk: NS - 1 slices stay queued while the host waits */
		if (bwb_hip_slot_upload(ctx, slot, w->params, c->fq.seq, c->fq.len, c->fq.n, c->fq.stride, c->carry, c->carry_len))
			bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
		if (j == 0) { /* the first chunk is staged, scratch and pool exist: now the index must be complete */
			if (bwb_hip_ctx_index_wait(ctx, &w->t_ctx)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
			uint64_t pb = 0;
			bwb_hip_setup_times(ctx, NULL, &w->t_pool, &pb);
			w->pool_bytes = pb;
			if (dbg) fprintf(stderr, "[bwb host] worker %d (device %d): index upload %.3f s, chunk pool %.1f GB allocated in %.3f s, first chunk staged at +%.3f s\n", w->gpu, w->device, w->t_ctx,
			                 (double)pb / (1u << 30), w->t_pool, wall() - tq);
		}
		if (bwb_hip_slot_submit(ctx, slot))
			bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
		if (j == 0) w->t_first_submit = wall() - tq;
		free(c->fq.seq); free(c->fq.len); free(c->carry); c->fq.seq = NULL; c->fq.len = NULL; c->carry = NULL; /* (staged by the library: the caller's buffers are free) */
		in_slot[slot] = c;
		w->n_reads += c->fq.n;
		j++;
		if (dbg) fprintf(stderr, "[bwb host] worker %d: chunk %zu (%u reads) submitted at +%.3f s\n", w->gpu, c->idx, c->fq.n, wall() - tq);
	}
	bwb_stats st;
	if (bwb_hip_flush(ctx) || bwb_hip_get_stats(ctx, &st)) bwb_die("align_reads_inexact_gpu: GPU %d: %s", w->device, bwb_hip_last_error());
	w->total = st;
	w->kernel_ms = st.ms_calc_d + st.ms_search;
	bwb_hip_ctx_destroy(ctx);
	if (dbg) fprintf(stderr, "[bwb host] worker %d: done %.3f s\n", w->gpu, wall() - tq);
	return NULL;
}

/* The GPU replacement for align_reads_inexact_parallel, as a pipeline: reader thread (FASTQ -> chunks) | one worker per GPU | this
 * thread (ordered writer, inexact_match.c:154-162).  `reads` is the FASTQ's NAME: the file is streamed, not loaded (the reference's
 * reads_t holds the whole file in memory before the first read is aligned). */
int align_reads_inexact_gpu_stream(bwt_t *BWT, const char *readsFname, aln_params_t *params, char *alnFname, int n_gpus) {
	printf("BWBBLE Inexact Alignment (MI355X)...\n");
	FILE *alnFile = fopen(alnFname, "a+b");                                   /* inexact_match.c:94 */
	if (!alnFile) { perror(alnFname); bwb_die("align_reads_inexact: Cannot open ALN file: %s!", alnFname); }
	if (bwb_hip_abi_version() != BWB_HIP_ABI_VERSION) bwb_die("align_reads_inexact_gpu: libbwbble_hip.so implements C-ABI version %d, this binary was compiled against %d", bwb_hip_abi_version(), BWB_HIP_ABI_VERSION);
	const int ndev = bwb_hip_device_count();
	if (ndev < 1) bwb_die("align_reads_inexact_gpu: no HIP device found (this build has no CPU alignment path)");
	/* BWB_DEVICE_MAP=d0,d1,...: worker g drives HIP device d_g (default g).  Several workers on one device are allowed - that
	 * is how the multi-worker path (shared queue, ordered writer) is tested on a one-GPU box. */
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
	pipe_t pp;
	memset(&pp, 0, sizeof(pp));
	pthread_mutex_init(&pp.mu, NULL);
	pthread_cond_init(&pp.cv_work, NULL); pthread_cond_init(&pp.cv_done, NULL); pthread_cond_init(&pp.cv_space, NULL);
	pp.readsFname = readsFname; pp.params = params;
	pp.chunk_reads = GPU_CHUNK_DEFAULT;
	if (getenv("BWB_CHUNK")) pp.chunk_reads = (uint32_t)strtoul(getenv("BWB_CHUNK"), NULL, 10);
	if (pp.chunk_reads < 1) pp.chunk_reads = 1;
	pp.max_ahead = (size_t)n_gpus * (BWB_MAX_SLOTS + 2); /* chunks parsed beyond the one being written: every worker's slots and two to take */
	worker_t *ws = (worker_t *)calloc((size_t)n_gpus, sizeof(worker_t));
	pthread_t *th = (pthread_t *)calloc((size_t)n_gpus, sizeof(pthread_t));
	pthread_t rth;
	const double t0 = wall();
	if (pthread_create(&rth, NULL, reader_thread, &pp)) bwb_die("align_reads_inexact_gpu: cannot start the reader thread");
	for (int g = 0; g < n_gpus; g++) {
		ws[g] = (worker_t){ .gpu = g, .device = nmap ? devmap[g] : g, .BWT = BWT, .params = params, .pp = &pp };
		if (pthread_create(&th[g], NULL, gpu_worker, &ws[g])) bwb_die("align_reads_inexact_gpu: cannot start a host thread");
	}
	/* ordered writer (the reference writes after each batch, inexact_match.c:154-162), woken when a chunk's hits have arrived */
	size_t processed = 0;
	const int dbg = getenv("BWB_DEBUG") != NULL;
	for (;;) {
		pthread_mutex_lock(&pp.mu);
		while (!(pp.first && pp.first->ready) && !(pp.reader_done && !pp.first)) pthread_cond_wait(&pp.cv_done, &pp.mu);
		chunk_t *c = pp.first;
		if (c) { pp.first = c->next; if (!pp.first) pp.last = NULL; }
		pthread_mutex_unlock(&pp.mu);
		if (!c) break;
		if (c->buf_len && fwrite(c->buf, 1, c->buf_len, alnFile) != c->buf_len) bwb_die("align_reads_inexact: Cannot write to the ALN file: %s!", alnFname);
		processed += c->n;
		printf("Processed %zu reads. Elapsed: %.2f sec\n", processed, wall() - t0);
		if (dbg) fprintf(stderr, "[bwb host] writer: chunk %zu written at +%.3f s\n", c->idx, wall() - t0);
		free(c->buf); free(c);
		pthread_mutex_lock(&pp.mu);
		pp.n_written++;
		pthread_cond_broadcast(&pp.cv_space);
		pthread_mutex_unlock(&pp.mu);
	}
	pthread_join(rth, NULL);
	bwb_stats tot; memset(&tot, 0, sizeof(tot));
	double kms = 0, tctx = 0, tpool = 0, tfirst = 0;
	unsigned long long pool_bytes = 0;
	for (int g = 0; g < n_gpus; g++) {
		pthread_join(th[g], NULL);
		tot.visits_single += ws[g].total.visits_single; tot.visits_alphabet += ws[g].total.visits_alphabet;
		tot.heap_pops += ws[g].total.heap_pops; tot.n_alignments += ws[g].total.n_alignments; tot.n_overflow_reads += ws[g].total.n_overflow_reads;
		if (ws[g].kernel_ms > kms) kms = ws[g].kernel_ms;
		if (ws[g].t_ctx > tctx) tctx = ws[g].t_ctx;
		if (ws[g].t_pool > tpool) tpool = ws[g].t_pool;
		if (ws[g].t_first_submit > tfirst) tfirst = ws[g].t_first_submit;
		if (ws[g].pool_bytes > pool_bytes) pool_bytes = ws[g].pool_bytes;
		if (n_gpus > 1) /* one line per worker: an uneven node (a slow link, a busy socket) shows here, not in the total */
			printf("  GPU %d (device %d, NUMA node %d): reads %llu  kernel %.1f ms  index to HBM %.2f sec  launches %llu  parked reads %llu  re-run reads %llu\n", g, ws[g].device,
			       bwb_hip_device_numa_node(ws[g].device), (unsigned long long)ws[g].n_reads, ws[g].kernel_ms, ws[g].t_ctx,
			       (unsigned long long)ws[g].total.launches_search, (unsigned long long)ws[g].total.n_parked_reads, (unsigned long long)ws[g].total.n_overflow_reads);
	}
	const double dt = wall() - t0;
	printf("GPUs: %d  reads: %llu  wall: %.3f sec (%.0f reads/s incl. index upload)  kernel: %.1f ms  index to HBM: %.2f sec  first chunk parsed after: %.2f sec  rank-block visits: %llu  hits: %llu  re-run reads: %llu\n",
	       n_gpus, (unsigned long long)pp.n_reads, dt, pp.n_reads / (dt > 0 ? dt : 1), kms, tctx, pp.t_first_chunk, (unsigned long long)(tot.visits_single + tot.visits_alphabet),
	       (unsigned long long)tot.n_alignments, (unsigned long long)tot.n_overflow_reads);
	/* where the start-up went (the three overlap): the .bwt file in memory | the index in HBM | the chunk pool's hipMalloc | first slice queued */
	printf("start-up: .bwt read %.2f sec | index to HBM %.2f sec | chunk pool %.1f GB in %.2f sec | first slice queued after %.2f sec\n",
	       bwt_load_seconds(BWT), tctx, (double)pool_bytes / (1u << 30), tpool, tfirst);
	free(ws); free(th);
	pthread_mutex_destroy(&pp.mu); pthread_cond_destroy(&pp.cv_work); pthread_cond_destroy(&pp.cv_done); pthread_cond_destroy(&pp.cv_space);
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
	/* The reference loads the index, then all reads, then aligns (align.c:50-76).  Here the three overlap: loader threads read the .bwt
	 * file while every GPU worker already uploads its head (bwt_io.c, bwb_hip_ctx_create_streamed), and the reader thread parses the
	 * FASTQ chunk by chunk while the GPUs align the chunks before. */
	bwt_t *BWT = load_bwt_start(bwtFname, 0);
	if (params->use_precalc) { /* align.c:59-65: the first `align -P` on an index leaves <fasta>.pre behind (precalc.c: written, never read) */
		char *preFname = (char *)malloc(L);
		snprintf(preFname, L, "%s.pre", fastaFname);
		FILE *pf = fopen(preFname, "r");
		if (pf) { /* load_precalc_sa_intervals (align.c:226-238) would read it: here only its shape is checked (precalc.c) */
			fclose(pf);
			if (check_precalc_file(preFname)) fprintf(stderr, "warning: %s is not a complete table of 16777216 interval lists (a run of the reference would fail on it): delete it to have it rebuilt\n", preFname);
		} else {
			load_bwt_wait(BWT);
			printf("Total BWT loading time: %.2f sec\n", wall() - t);
			t = wall();
			precalc_sa_intervals(BWT, params, preFname);
			printf("Total pre-calculated intervals time: %.2f sec\n", wall() - t);
		}
		free(preFname);
	}
	t = wall();
	align_reads_inexact_gpu_stream(BWT, readsFname, params, alnsFname, n_gpus);   /* the seam: align.c:72-76 */
	printf("Total read alignment time (index and read loading overlapped): %.2f sec\n", wall() - t);
	free_bwt(BWT);
	free(bwtFname);
	return 0;
}
