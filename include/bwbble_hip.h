/*
 * bwbble_hip.h - C-ABI of the MI355X (gfx950) read-alignment library, libbwbble_hip.so.
 *
 * This is the drop-in boundary for BWBBLE's `align` hot path.  The reference has no plugin/FFI
 * layer; its seam is the function pair selected in align_reads (mg-aligner/align.c:72-76):
 *
 *     int align_reads_inexact[_parallel](bwt_t*, reads_t*, sa_intv_list_t*, aln_params_t*, char* alnFname);
 *                                                               (mg-aligner/inexact_match.h:39-40)
 *
 * A maintainer adds `align_reads_inexact_gpu()` with that same signature next to them (see
 * INTEGRATION.md and bwbble_amd/host/align_gpu.c) and it calls the entry points below.  Plain
 * pointers and sizes only; every function returns 0 on success or a negative BWB_E_* code, never
 * exits the process (the reference printf+exit(1)s; the host wrapper keeps that behaviour), and
 * bwb_hip_last_error() gives the message.  Calls on distinct contexts are thread-safe (one host
 * thread per GPU); a context must not be used from two threads at once.
 */
#ifndef BWBBLE_HIP_H
#define BWBBLE_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BWB_OK 0
#define BWB_E_ARG (-1)         /* bad argument / unsupported parameter combination */
#define BWB_E_HIP (-2)         /* HIP runtime error (no device, out of memory, launch failure) */
#define BWB_E_OVERFLOW (-3)    /* a read exceeded the largest per-read scratch class */
#define BWB_E_STATE (-4)       /* call order violated (e.g. run before upload) */

/* mirror of aln_params_t (mg-aligner/align.h:48-79): same fields, same order, same meaning.
 * Defaults are set_default_aln_params (mg-aligner/align.c:22-38). n_threads is ignored by the GPU. */
typedef struct {
	int32_t max_diff, max_gapo, max_gape, max_entries;
	int32_t mm_score, gapo_score, gape_score;
	int32_t seed_length, max_diff_seed, max_best, no_indel_length;
	int32_t matched_Ncontig, use_precalc, is_multiref, n_threads;
} bwb_params;

/* One alignment hit = one aln_t (mg-aligner/align.h:81-90) in fixed 48-byte form (ABI version 3; 32 bytes with four runs before).
 * The edit path is all STATE_M except for <= 8 gap runs (one per gap open: -o up to 8):
 * run = start_index_in_path | (run_length << 8) | (is_deletion << 15); unused runs are 0xFFFF. */
#define BWB_MAX_GAP_RUNS 8
typedef struct {
	uint64_t L, U;           /* SA interval */
	uint16_t score;          /* aln_t.score: the full int aln_score(num_mm, num_gapo, num_gape) (inexact_match.c:332,348); up to 1 024 heap
	                            buckets are accepted, so it does not fit the 8 bits it had in ABI version 1 */
	uint8_t num_mm, num_gapo, num_gape, reserved;
	uint16_t aln_length;     /* 8-bit wrapped like aln_entry_t.aln_length (align.h:104) */
	uint16_t gap_run[BWB_MAX_GAP_RUNS];
	uint64_t reserved2;      /* (zero; the record is three 16-byte words) */
} bwb_aln;

/* Results of one batch: read r owns alns[aln_off[r] .. aln_off[r+1]) in discovery order
 * (== the order of alns->entries in the reference, align.c:286-297). Owned by the context;
 * valid until the next bwb_hip_batch_* call on it. */
typedef struct {
	uint32_t n_reads;
	const uint64_t *aln_off;     /* n_reads + 1 */
	const bwb_aln *alns;
} bwb_result;

/* Work/timing counters since the last bwb_hip_reset_stats / bwb_hip_batch_run (SURVEY.md 8(d) counting rules) */
typedef struct {
	uint64_t visits_single;      /* rank-block visits made for the 7-code exact steps (calculate_d, exact tail) */
	uint64_t visits_calc_d;      /* the part of visits_single made by the calculate_d kernel */
	uint64_t visits_alphabet;    /* rank-block visits made for O_alphabet (2 per expansion unless special-cased) */
	uint64_t heap_pops, heap_pushes;
	uint64_t n_alignments;
	uint64_t n_overflow_reads;   /* reads re-run with a larger scratch class (still on the GPU) */
	uint64_t n_parked_reads;     /* reads that were parked at the end of a slice and resumed by the next launch */
	uint64_t bucket_loads_search; /* 128-byte device buckets the search kernel fetched (a same-bucket L-1/U pair counts once) */
	uint64_t bucket_loads_calc_d; /* the same for the calculate_d kernel */
	uint64_t lane_iterations, wave_iterations; /* search loop iterations of busy lanes / of waves: their ratio = lanes busy of 64 */
	uint64_t heap_entries_stored, heap_entries_loaded; /* heap entries the search kernel wrote to / read from its chunk pool (deletion children
	                                                      travel as one group entry and a match child that is popped next stays in registers: fewer than pushes) */
	uint64_t record_loads;       /* 16-byte records (four read positions each: D, D_seed, bases) the search kernel loaded */
	double ms_calc_d;            /* HIP-event time of the calculate_d kernel(s) that have finished */
	double ms_search;            /* HIP-event time of the inexact-search kernel(s) that have finished, all passes */
	double ms_total;             /* batch_run: wall time of the call */
	uint32_t launches_calc_d, launches_search;
} bwb_stats;

typedef struct bwb_hip_ctx bwb_hip_ctx;

/* Version of this interface: 3 since bwb_aln holds eight gap runs in 48 bytes (round 5; 2: bwb_aln.score 16 bits wide, round 4).  A binding compiled against another version must not
 * be used with the library: compare BWB_HIP_ABI_VERSION with bwb_hip_abi_version() at start-up. */
#define BWB_HIP_ABI_VERSION 3
int bwb_hip_abi_version(void);

int bwb_hip_device_count(void);
int bwb_hip_device_numa_node(int device);                    /* NUMA node of the device's PCIe root, -1 = unknown / single node */
const char *bwb_hip_last_error(void);
void bwb_default_params(bwb_params *p);                       /* align.c:22-38 */

/* Creates a context on `device` and builds the device FM-index from the reference's in-memory
 * bwt_t arrays (mg-aligner/bwt.h:19-40, file layout bwt.c:66-82):
 *   hdr = {length, num_words, num_sa, num_occ, sa0_index}, C[17], bwt[num_words], O[num_occ*16].
 * The index is re-laid-out on the GPU into 128-byte rank buckets of 64 BWT characters (2 bytes of device memory per BWT
 * character); the host arrays are not referenced after return.  One context per device is the intended use: with the first batch a context sizes its heap
 * chunk pool from what the device has free (minus a reserve for the re-run classes and further slots), so a second context
 * on the same device - tests do that - should be given a budget with the environment variable BWB_POOL_GB. */
int bwb_hip_ctx_create(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                       const uint64_t *O, bwb_hip_ctx **out);
/* The same while the host arrays are still being filled (`bwbble align` reads the 12 GB .bwt of a GRCh37-scale index with several
 * threads while the GPUs already take it in): *blocks_ready = number of leading 128-character blocks whose bwt words [16 k, 16 k + 16)
 * and O rows k are in memory, advanced by the caller's loader with release semantics; the upload of a chunk waits for it.  NULL =
 * everything is there (= bwb_hip_ctx_create).  Several contexts (one per GPU) may follow the same counter. */
int bwb_hip_ctx_create_streamed(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                                const uint64_t *O, const volatile uint64_t *blocks_ready, bwb_hip_ctx **out);
/* The same, returning as soon as the context exists: the index upload runs on a thread of the library, so the caller can already
 * slot_upload its first batch - which also sizes and allocates the context's scratch and heap chunk pool, seconds of hipMalloc at GRCh37
 * scale - while the index is still on its way.  The host arrays (and *blocks_ready) must stay valid until bwb_hip_ctx_index_wait has
 * returned; every entry point that launches a kernel waits by itself (slot_submit, batch_run, calc_d, rank16, rank_bench, locate), and so
 * does ctx_destroy.  index_wait reports an upload error (the message through bwb_hip_last_error) and, optionally, the upload's seconds.
 * (Added in round 6 without a version change: no structure or existing entry point changed.) */
int bwb_hip_ctx_create_async(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                             const uint64_t *O, const volatile uint64_t *blocks_ready, bwb_hip_ctx **out);
int bwb_hip_ctx_index_wait(bwb_hip_ctx *ctx, double *seconds);
/* where a context's start-up time went: seconds of the index upload (-1 while it is running), seconds spent in the hipMalloc of the heap
 * chunk pool and the pool's size (any pointer may be NULL) */
int bwb_hip_setup_times(bwb_hip_ctx *ctx, double *index_seconds, double *pool_seconds, uint64_t *pool_bytes);
/* The calculate_d table of the context: the state of calculate_d (inexact_match.c:171-254) after its first K steps for every
 * K-mer, from which kl_calc_d starts a read and its seed.  Built by the first slot_submit / batch_run whose batch has at least 200 000 reads
 * (environment: BWB_DTAB=1 always, 0 never; BWB_DTAB_K: K, default and maximum 12); K = 0: none.  Results do not depend on it. */
int bwb_hip_dtab_info(bwb_hip_ctx *ctx, int *K, double *build_seconds, uint64_t *bytes);
void bwb_hip_ctx_destroy(bwb_hip_ctx *ctx);

/* Replaces align_reads_inexact[_parallel] for one batch (inexact_match.c:25-168): calculate_d x2 +
 * inexact_match for n_reads reads. reads_fwd holds read->seq codes (A0 G1 C2 T3 N4, io.h:112-130),
 * `stride` bytes per read; the reverse complement (read->rc, io.c:502-504) is formed on the GPU.
 * == batch_upload + batch_run + batch_result. */
int bwb_hip_align_batch(bwb_hip_ctx *ctx, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                        uint32_t n_reads, uint32_t stride, bwb_result *out);

/* The same in three steps, so that a caller can time the GPU work with inputs resident in HBM. */
int bwb_hip_batch_upload(bwb_hip_ctx *ctx, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                         uint32_t n_reads, uint32_t stride);
int bwb_hip_batch_run(bwb_hip_ctx *ctx);                       /* kernels only; blocks until done; resets the statistics first */
int bwb_hip_batch_result(bwb_hip_ctx *ctx, bwb_result *out);  /* D2H copy of the hits */
int bwb_hip_get_stats(bwb_hip_ctx *ctx, bwb_stats *out);
int bwb_hip_reset_stats(bwb_hip_ctx *ctx);

/* Streaming form of the same (what `bwbble align` uses for a FASTQ of many batches, inexact_match.c:103-165): up to
 * BWB_MAX_SLOTS batches are resident per context.  slot_upload copies the reads to HBM on a copy stream (the caller's
 * buffers are free when it returns); slot_submit queues calculate_d and one slice of the search for that slot and returns at
 * once - a slice does not drain: the reads still under way when the slot's cursor runs out are parked and resumed by the
 * next submitted slot's slice, so the heavy tail of one batch overlaps the bulk of the next; slot_wait blocks until every
 * read of the slot is done (launching a draining slice if nothing else is queued); slot_result = slot_wait + D2H of the
 * hits on a result stream, valid until the slot is uploaded again; flush = wait for every slot.  All slots in flight use
 * the same parameters (a slot_upload with different ones flushes first).  Same result bytes as align_batch.
 *
 * Reads that are not longer than the seed (len <= seed_length): the reference computes D_seed only for longer reads
 * (inexact_match.c:62-64) yet inexact_match consults it for every read (:321-328), so its serial path (-t 1) gives a short
 * read the bounds of the last longer read before it in the file (zeros if there is none; with -t N it depends on the thread
 * that happens to process the read).  The library reproduces the serial behaviour: inside a batch it knows the order; for
 * the head of a batch the caller passes that last longer read of the earlier batches as carry_seq/carry_len (read->seq
 * codes; NULL/0 = none, which is also what batch_upload/align_batch assume). */
#define BWB_MAX_SLOTS 8
int bwb_hip_slot_upload(bwb_hip_ctx *ctx, int slot, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                        uint32_t n_reads, uint32_t stride, const uint8_t *carry_seq, uint32_t carry_len);
int bwb_hip_slot_submit(bwb_hip_ctx *ctx, int slot);
int bwb_hip_slot_wait(bwb_hip_ctx *ctx, int slot);
int bwb_hip_slot_result(bwb_hip_ctx *ctx, int slot, bwb_result *out);
int bwb_hip_flush(bwb_hip_ctx *ctx);

/* calculate_d for the uploaded batch (inexact_match.c:171-254): D and D_seed of every read as
 * (num_diff, sa_intv_width) int32 pairs, out_D[n_reads][max_len+1][2], out_Dseed[n_reads][seed_length+1][2]
 * (rows beyond a read's length are 0; D_seed rows are 0 when len <= seed_length). For parity tests. */
int bwb_hip_calc_d(bwb_hip_ctx *ctx, int32_t *out_D, int32_t *out_Dseed);

/* O_alphabet (bwt.c:374-438) for n positions: out[q][j] = C[j] + Occ(j, pos[q]) + inc with the
 * reference's three-base-code behaviour when exact==0, or the exact count for all 15 codes
 * (== C[j] + O(j,pos) + inc, bwt.c:348-372) when exact!=0. out[q][0] is 0. pos may be (uint64_t)-1. */
int bwb_hip_rank16(bwb_hip_ctx *ctx, const uint64_t *pos, size_t n, int inc, int exact, uint64_t *out);

/* Rank micro-benchmark: `n` pseudo-random Occ16 queries (seeded), repeated `iters` times with
 * everything resident; returns kernel milliseconds per iteration and a checksum of the results. */
int bwb_hip_rank_bench(bwb_hip_ctx *ctx, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum);
/* The same queries with the layout the alignment kernels use: one query per lane, every lane gathers its own 128-byte
 * bucket and ranks all 15 codes (the octet version above splits one bucket over 8 lanes). Same checksum. */
int bwb_hip_rank_bench_lane(bwb_hip_ctx *ctx, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum);

/* SA[i] for n suffix-array rows via the invPsi walk (bwt.c:311-329); needs the sampled SA
 * (bwt.c:80) uploaded with bwb_hip_set_sa. Used by aln2sam (align.c:760-812). */
int bwb_hip_set_sa(bwb_hip_ctx *ctx, const uint64_t *SA, uint64_t num_sa);
int bwb_hip_locate(bwb_hip_ctx *ctx, const uint64_t *rows, size_t n, uint64_t *out_pos);
/* the last bwb_hip_locate call on the context: rows looked up, invPsi steps taken (each one rank-block visit: a 128-byte bucket) and the
 * kernel's HIP-event time - what `bwbble aln2sam` and bench.py report (any pointer may be NULL) */
int bwb_hip_locate_stats(bwb_hip_ctx *ctx, uint64_t *rows, uint64_t *steps, double *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif
