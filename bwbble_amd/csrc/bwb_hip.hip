/*
 * bwb_hip.hip - C-ABI implementation of include/bwbble_hip.h (libbwbble_hip.so), MI355X / gfx950 only.
 *
 * Host orchestration of one batch (replaces the per-batch body of align_reads_inexact_parallel,
 * mg-aligner/inexact_match.c:103-165):
 *   1. kl_calc_d over all reads (one read per lane, lanes pull reads from a global cursor);
 *   2. kl_search over all reads with the class-0 per-lane scratch and the shared heap chunk pool; reads whose
 *      interval list / hit list did not fit, or that found the pool empty, are re-run -- still on the GPU --
 *      in class 1, then class 2 (fewer lanes, larger lists, a fresh pool);
 *   3. hits gathered into read order.
 * There is no CPU fallback anywhere in this file.
 */
#include <hip/hip_runtime.h>
#include <time.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/bwbble_hip.h"
#include "bwb_kernels.h"
static double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define DBG_T(what) do { if (getenv("BWB_DEBUG_T")) { hipDeviceSynchronize(); fprintf(stderr, "[bwb t] %s: %.3f s\n", what, wall_s() - dbg_t0); dbg_t0 = wall_s(); } } while (0)
#include "bwb_lane.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &m) { g_err = m; return code; }
#define HIPCHK(x)                                                                                         \
	do {                                                                                                  \
		hipError_t e_ = (x);                                                                              \
		if (e_ != hipSuccess) return fail(BWB_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
	} while (0)

/* per-lane scratch of one class (class 0 = every resident lane, classes 1/2 = fewer lanes with larger lists) */
struct ScratchClass {
	void *mem = nullptr;
	size_t bytes = 0;
	LaneScratch sc{};
	uint32_t blocks = 0;
};

struct bwb_hip_ctx {
	int device = 0, num_cu = 0;
	hipStream_t stream = nullptr;
	DevIndex ix{};
	uint4 *d_buckets = nullptr;
	uint64_t sa0_index = 0, num_sa = 0;
	uint64_t *d_SA = nullptr;
	bool pos32 = true;                  /* BWT rows fit 32-bit positions */
	/* batch */
	bool uploaded = false, ran = false;
	bwb_params p{};
	KParams kp{};
	uint32_t n_reads = 0, stride = 0, maxlen = 0;
	size_t cap_reads = 0, cap_readbytes = 0, cap_dbuf = 0;
	uint8_t *d_reads = nullptr, *d_dbuf = nullptr, *d_status = nullptr;
	uint16_t *d_lens = nullptr;
	uint32_t *d_counter = nullptr, *d_worklist = nullptr, *d_n = nullptr;
	uint64_t *d_off = nullptr, *d_dstoff = nullptr;
	uint4 *d_log = nullptr, *d_sorted = nullptr;
	unsigned long long *d_count = nullptr, *d_stats = nullptr;
	uint64_t log_cap = 0, sorted_cap = 0;
	uint32_t dstride = 0;
	ScratchClass cls[3];
	uint4 *d_pool = nullptr;            /* heap chunk pool shared by all lanes and classes, POOL_REGIONS equal regions */
	size_t pool_bytes = 0;
	unsigned int *d_pool_bump = nullptr;
	uint32_t keep = 256;                /* chunks of a lane's private run (BWB_KEEP) */
	uint32_t *d_dbg_iters = nullptr;    /* BWB_DEBUG_ITERS: per-read iteration counts */
	int bpc_search = 2, bpc_calcd = 2;
	bool wide = false;                  /* 32-byte heap entries (max_gapo > 1: more than one gap run per path) */
	std::vector<uint8_t> h_status;
	std::vector<uint64_t> h_aln_off;
	std::vector<bwb_aln> h_alns;
	bwb_stats stats{};
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

extern "C" const char *bwb_hip_last_error(void) { return g_err.c_str(); }

extern "C" int bwb_hip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" void bwb_default_params(bwb_params *p) { /* set_default_aln_params, align.c:22-38 */
	memset(p, 0, sizeof(*p));
	p->gape_score = 4; p->gapo_score = 11; p->mm_score = 3; p->max_diff = 0; p->max_gape = 6; p->max_gapo = 1;
	p->seed_length = 32; p->max_diff_seed = 2; p->max_entries = 3000000; p->use_precalc = 0; p->matched_Ncontig = 0;
	p->is_multiref = 1; p->max_best = 30; p->no_indel_length = 5; p->n_threads = 1;
}

extern "C" int bwb_hip_ctx_create(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                                  const uint64_t *O, bwb_hip_ctx **out) {
	if (!hdr || !C || !bwt || !O || !out) return fail(BWB_E_ARG, "ctx_create: null argument");
	const uint64_t length = hdr[0], num_words = hdr[1], num_occ = hdr[3];
	const uint64_t nblk = (length + 127) / 128;
	if (length < 2 || num_occ != nblk || num_words != (length + 7) / 8) return fail(BWB_E_ARG, "ctx_create: inconsistent .bwt header");
	const uint64_t nsb = (nblk + (1ull << BWB_SB_SHIFT) - 1) >> BWB_SB_SHIFT;
	if (nsb > BWB_NSB_MAX) return fail(BWB_E_ARG, "ctx_create: index larger than 2^34 characters");
	HIPCHK(hipSetDevice(device));
	bwb_hip_ctx *c = new bwb_hip_ctx();
	c->device = device;
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	c->num_cu = prop.multiProcessorCount;
	HIPCHK(hipStreamCreate(&c->stream));
	HIPCHK(hipEventCreate(&c->ev0));
	HIPCHK(hipEventCreate(&c->ev1));
	HIPCHK(hipMalloc(&c->d_buckets, nblk * 128));
	HIPCHK(hipMalloc(&c->d_counter, 64));
	HIPCHK(hipMalloc(&c->d_count, 64));
	HIPCHK(hipMalloc(&c->d_stats, sizeof(unsigned long long) * 40));

	/* superblock base table */
	std::vector<uint64_t> sbcount(BWB_NSB_MAX * 16, 0);
	memset(&c->ix, 0, sizeof(c->ix));
	for (uint64_t sb = 0; sb < nsb; sb++) {
		const uint64_t blk = sb << BWB_SB_SHIFT;
		const uint32_t first = bwt[blk * 16] >> 28;
		for (int j = 0; j < 16; j++) sbcount[sb * 16 + j] = O[blk * 16 + j] - ((first == (uint32_t)j && !(j == 0 && blk * 128 == hdr[4])) ? 1 : 0);
	}
	for (uint64_t sb = 0; sb < BWB_NSB_MAX; sb++)
		for (int j = 0; j < 16; j++) c->ix.base[sb][j] = C[j] + sbcount[sb * 16 + j];
	for (int j = 0; j < 16; j++) { c->ix.base[BWB_ROW_NEG][j] = C[j]; c->ix.base[BWB_ROW_END][j] = C[j + 1]; }
	c->ix.buckets = c->d_buckets;
	c->ix.length = length;
	c->ix.nblk = nblk;
	c->sa0_index = hdr[4];
	c->pos32 = length < 0xFFFFFFFFull && !getenv("BWB_FORCE_POS64");

	/* re-layout on the GPU, 2^20 blocks (128 M characters) per chunk */
	const uint64_t CH = 1ull << 20;
	uint32_t *d_bwt = nullptr; uint64_t *d_O = nullptr, *d_sbc = nullptr;
	HIPCHK(hipMalloc(&d_bwt, std::min(CH, nblk) * 64));
	HIPCHK(hipMalloc(&d_O, std::min(CH, nblk) * 128));
	HIPCHK(hipMalloc(&d_sbc, sbcount.size() * 8));
	HIPCHK(hipMemcpy(d_sbc, sbcount.data(), sbcount.size() * 8, hipMemcpyHostToDevice));
	for (uint64_t b0 = 0; b0 < nblk; b0 += CH) {
		const uint64_t nb = std::min(CH, nblk - b0);
		const uint64_t w0 = b0 * 16, nw = std::min(nb * 16, num_words - w0);
		HIPCHK(hipMemcpy(d_bwt, bwt + w0, nw * 4, hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(d_O, O + b0 * 16, nb * 128, hipMemcpyHostToDevice));
		const uint64_t nthreads = nb * 8;
		hipLaunchKernelGGL(k_relayout, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, d_bwt, d_O, b0, nb, nw, hdr[4], d_sbc, c->d_buckets);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	hipFree(d_bwt); hipFree(d_O); hipFree(d_sbc);
	*out = c;
	return BWB_OK;
}

static void free_batch(bwb_hip_ctx *c) {
	hipFree(c->d_reads); hipFree(c->d_lens); hipFree(c->d_dbuf); hipFree(c->d_status); hipFree(c->d_worklist);
	hipFree(c->d_n); hipFree(c->d_off); hipFree(c->d_dstoff);
	c->d_reads = c->d_dbuf = c->d_status = nullptr; c->d_lens = nullptr; c->d_worklist = c->d_n = nullptr; c->d_off = c->d_dstoff = nullptr;
	c->cap_reads = c->cap_readbytes = c->cap_dbuf = 0;
}

extern "C" void bwb_hip_ctx_destroy(bwb_hip_ctx *c) {
	if (!c) return;
	hipSetDevice(c->device);
	free_batch(c);
	for (auto &k : c->cls) hipFree(k.mem);
	hipFree(c->d_pool); hipFree(c->d_pool_bump);
	hipFree(c->d_log); hipFree(c->d_sorted); hipFree(c->d_buckets); hipFree(c->d_counter); hipFree(c->d_count);
	hipFree(c->d_stats); hipFree(c->d_SA);
	if (c->ev0) hipEventDestroy(c->ev0);
	if (c->ev1) hipEventDestroy(c->ev1);
	if (c->stream) hipStreamDestroy(c->stream);
	delete c;
}


static size_t lane_lds(const bwb_hip_ctx *c) {
	return (size_t)BWB_BASE_ROWS * 16 * 8 + (size_t)2 * KID_ROWS * LANE_BLOCK * (c->pos32 ? 4 : 8);
}

/* The heap chunk pool, sized for the batch at hand and grown when a larger batch or index needs more (the pool is empty
 * between launches, so it can be replaced at upload time).  hipMalloc costs about 27 ms per GB here: 4.7 s for the 172 GB
 * a GRCh37-scale batch uses, which a 600-read run should not pay. */
static int ensure_pool(bwb_hip_ctx *c) {
	size_t fr = 0, tot = 0;
	hipMemGetInfo(&fr, &tot);
	fr += c->pool_bytes; /* what would be free without the current pool */
	/* Ceiling: 60 % of what is free (the rest is for the scratch classes and the hit log), and what the regions can name:
	 * a state word holds a 26-bit chunk index relative to the block's region. */
	const size_t ceiling = std::min<size_t>(fr / 10 * 6, (size_t)POOL_REGIONS << 36);
	/* Need: per read in flight, the private run (keep chunks of 1 KB, 2 KB with 32-byte entries) plus a share of the
	 * common part that grows with the index: measured 37 KB per lane at 106 M rows, ~300 KB at 884 M, 870 KB at 6.85 G. */
	const size_t lanes = std::min<size_t>(std::max<uint32_t>(c->n_reads, 1), (size_t)c->num_cu * 2 * LANE_BLOCK);
	const size_t index_mb = (size_t)(c->ix.nblk >> 13) + 1;
	const size_t per_lane = ((size_t)c->keep << 10) + std::min<size_t>((size_t)1536 << 10, index_mb << 9);
	size_t want = std::max<size_t>((size_t)1 << 30, lanes * per_lane / 4 * 5 * (c->wide ? 2 : 1));
	if (want > ceiling) want = ceiling;
	if (getenv("BWB_POOL_GB")) want = std::min<size_t>((size_t)atol(getenv("BWB_POOL_GB")) << 30, fr / 10 * 7);
	if (want < ((size_t)64 << 20)) want = (size_t)64 << 20; /* floor (also what BWB_POOL_GB=0 selects, to test the re-run path) */
	want &= ~(size_t)(POOL_REGIONS * 4096 - 1);
	if (c->d_pool && c->pool_bytes >= want) return BWB_OK;
	if (c->d_pool) { HIPCHK(hipFree(c->d_pool)); c->d_pool = nullptr; c->pool_bytes = 0; }
	if (want > fr) return fail(BWB_E_HIP, "not enough device memory for the heap chunk pool");
	{ double dbg_t0 = wall_s(); HIPCHK(hipMalloc(&c->d_pool, want)); DBG_T("ensure_pool: hipMalloc of the pool"); }
	if (!c->d_pool_bump) HIPCHK(hipMalloc(&c->d_pool_bump, POOL_REGIONS * 64));
	c->pool_bytes = want;
	return BWB_OK;
}

static int ensure_class(bwb_hip_ctx *c, int k) {
	int rc = ensure_pool(c);
	if (rc) return rc;
	ScratchClass &s = c->cls[k];
	uint32_t blocks, lcap, acap;
	if (k == 0) {
		/* 2 blocks (8 waves) per CU for both kernels (a third kl_calc_d block per CU measured no faster) */
		c->bpc_search = 2; c->bpc_calcd = 2;
		if (getenv("BWB_BLOCKS_PER_CU")) c->bpc_search = std::max(1, atoi(getenv("BWB_BLOCKS_PER_CU")));
		if (getenv("BWB_KEEP")) c->keep = (uint32_t)std::max(0, atoi(getenv("BWB_KEEP")));
		if (getenv("BWB_CALCD_BLOCKS_PER_CU")) c->bpc_calcd = std::max(1, atoi(getenv("BWB_CALCD_BLOCKS_PER_CU")));
		blocks = (uint32_t)(c->num_cu * std::max(c->bpc_search, c->bpc_calcd)); lcap = 4096; acap = 256;
	} else if (k == 1) {
		blocks = (uint32_t)c->num_cu; lcap = 8192; acap = 1024;
	} else {
		blocks = 4; lcap = 1u << 20; acap = 1u << 16;
	}
	const uint32_t nslots = blocks * LANE_BLOCK;
	const uint32_t wstride = c->maxlen + 1;
	const size_t isz = c->pos32 ? 8 : 16;
	const size_t b_bstate = (size_t)BSTATE_ROW * nslots * 4, b_lists = (size_t)nslots * 2 * lcap * isz, b_alns = (size_t)nslots * acap * 32,
	             b_winfo = 0;
	auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const size_t bytes = al(b_bstate) + al(b_lists) + al(b_alns) + al(b_winfo);
	if (!(s.mem && s.bytes >= bytes)) {
		if (s.mem) { hipFree(s.mem); s.mem = nullptr; }
		size_t fr = 0, tot = 0;
		hipMemGetInfo(&fr, &tot);
		if (bytes + ((size_t)1 << 30) > fr) return fail(BWB_E_HIP, "not enough device memory for the per-lane scratch (class " + std::to_string(k) + ")");
		{ double dbg_t0 = wall_s(); HIPCHK(hipMalloc(&s.mem, bytes)); DBG_T("ensure_class: hipMalloc of the class scratch"); }
		s.bytes = bytes;
	}
	unsigned char *base = (unsigned char *)s.mem;
	s.sc.bstate = (uint32_t *)base; base += al(b_bstate);
	s.sc.lists = (void *)base; base += al(b_lists);
	s.sc.alns = (uint4 *)base; base += al(b_alns);
	s.sc.winfo = (uint2 *)base;
	s.sc.nslots = nslots; s.sc.lcap = lcap; s.sc.acap = acap; s.sc.wstride = wstride;
	s.sc.pool = c->d_pool; s.sc.pool_bump = c->d_pool_bump; s.sc.keep = c->keep;
	s.blocks = blocks;
	return BWB_OK;
}

extern "C" int bwb_hip_batch_upload(bwb_hip_ctx *c, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                                    uint32_t n_reads, uint32_t stride) {
	if (!c || !p || (n_reads && (!reads_fwd || !lens)) || stride == 0) return fail(BWB_E_ARG, "batch_upload: bad argument");
	if (p->max_gapo < 0 || p->max_gapo > 4) return fail(BWB_E_ARG, "max_gapo (-o) must be in [0,4] on the GPU path");
	if (p->max_gape < 0 || p->max_gape > 100 || p->max_diff < 0 || p->max_diff > 100) return fail(BWB_E_ARG, "max_gape/max_diff out of the supported range [0,100]");
	if (p->seed_length < 0 || p->seed_length > 255) return fail(BWB_E_ARG, "seed_length must be in [0,255]");
	if (p->mm_score < 0 || p->gapo_score < 0 || p->gape_score < 0) return fail(BWB_E_ARG, "negative penalties are not supported");
	const int nb = (p->max_diff + 1) * p->mm_score + (p->max_gapo + 1) * p->gapo_score + (p->max_gape + 1) * p->gape_score; /* heap_init :513 */
	if (nb < 1 || nb > 128) return fail(BWB_E_ARG, "score range (heap buckets) must be in [1,128]");
	if (p->max_entries < 1) return fail(BWB_E_ARG, "max_entries must be positive");
	uint32_t maxlen = 0;
	for (uint32_t i = 0; i < n_reads; i++) maxlen = std::max<uint32_t>(maxlen, lens[i]);
	if (p->use_precalc)
		for (uint32_t i = 0; i < n_reads; i++)
			if (lens[i] < PRECALC_LEN) return fail(BWB_E_ARG, "-P needs reads of at least 12 bases (read2index, align.c:174-186, reads before the buffer otherwise)");
	if (maxlen > 255 || maxlen > stride) return fail(BWB_E_ARG, "reads longer than 255 bases (or than stride) are not supported (aln_entry_t.i is 8-bit, align.h:104)");
	double dbg_t0 = wall_s();
	HIPCHK(hipSetDevice(c->device));
	DBG_T("upload: checks + setdevice");
	c->p = *p;
	c->kp = KParams{ p->max_diff, p->max_gapo, p->max_gape, p->max_entries, p->mm_score, p->gapo_score, p->gape_score,
	                 p->seed_length, p->max_diff_seed, p->max_best, p->no_indel_length, nb, p->use_precalc ? 1 : 0, p->is_multiref ? 1 : 0 };
	c->n_reads = n_reads; c->stride = stride; c->maxlen = maxlen;
	c->wide = p->max_gapo > 1;
	/* per read: an 8-byte record {D pair, D_seed pair, base} for i = 0..maxlen+1, then 16 bytes (work, N count) */
	c->dstride = 8 * (maxlen + 2) + 16;
	const size_t nr = n_reads ? n_reads : 1;
	if (nr > c->cap_reads || nr * stride > c->cap_readbytes || nr * c->dstride > c->cap_dbuf) {
		free_batch(c);
		c->cap_reads = nr; c->cap_readbytes = nr * stride; c->cap_dbuf = nr * c->dstride;
		HIPCHK(hipMalloc(&c->d_reads, c->cap_readbytes));
		HIPCHK(hipMalloc(&c->d_lens, nr * 2));
		HIPCHK(hipMalloc(&c->d_dbuf, c->cap_dbuf));
		HIPCHK(hipMalloc(&c->d_status, nr));
		HIPCHK(hipMalloc(&c->d_worklist, nr * 4));
		HIPCHK(hipMalloc(&c->d_n, nr * 4));
		HIPCHK(hipMalloc(&c->d_off, nr * 8));
		HIPCHK(hipMalloc(&c->d_dstoff, nr * 8));
	}
	if (n_reads) {
		HIPCHK(hipMemcpyAsync(c->d_reads, reads_fwd, (size_t)n_reads * stride, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(c->d_lens, lens, (size_t)n_reads * 2, hipMemcpyHostToDevice, c->stream));
	}
	DBG_T("upload: batch buffers + copies");
	/* hit log: start at 8 records per read, grown on demand */
	const uint64_t want = std::max<uint64_t>((uint64_t)nr * 8, 1u << 16);
	if (c->log_cap < want) {
		hipFree(c->d_log);
		c->d_log = nullptr;
		HIPCHK(hipMalloc(&c->d_log, want * 32));
		c->log_cap = want;
	}
	if (getenv("BWB_DEBUG_ITERS")) { hipFree(c->d_dbg_iters); c->d_dbg_iters = nullptr; HIPCHK(hipMalloc(&c->d_dbg_iters, nr * 4)); HIPCHK(hipMemset(c->d_dbg_iters, 0, nr * 4)); }
	DBG_T("upload: hit log");
	int rc = ensure_class(c, 0);
	if (rc) return rc;
	DBG_T("upload: ensure_class (pool + scratch)");
	HIPCHK(hipStreamSynchronize(c->stream));
	c->uploaded = true; c->ran = false;
	return BWB_OK;
}

static Batch make_batch(bwb_hip_ctx *c, const uint32_t *worklist, uint32_t n_work) {
	Batch b;
	b.reads = c->d_reads; b.lens = c->d_lens; b.n_reads = c->n_reads; b.stride = c->stride;
	b.dbuf = c->d_dbuf; b.dstride = c->dstride;
	b.worklist = worklist; b.n_work = n_work; b.counter = c->d_counter; b.status = c->d_status;
	b.dbg_iters = c->d_dbg_iters;
	b.iter_budget = 0; b.lane_stride = 1;
	return b;
}

/* reads whose status == want -> device worklist; returns count */
static int collect(bwb_hip_ctx *c, uint8_t want, std::vector<uint32_t> &ids) {
	c->h_status.resize(c->n_reads);
	HIPCHK(hipMemcpy(c->h_status.data(), c->d_status, c->n_reads, hipMemcpyDeviceToHost));
	ids.clear();
	for (uint32_t i = 0; i < c->n_reads; i++) if (c->h_status[i] == want) ids.push_back(i);
	if (!ids.empty()) HIPCHK(hipMemcpy(c->d_worklist, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
	return BWB_OK;
}

static int launch_calc_d(bwb_hip_ctx *c, int k, const uint32_t *wl, uint32_t n_work, int32_t *dbgD, int32_t *dbgDs) {
	ScratchClass &s = c->cls[k];
	Batch b = make_batch(c, wl, n_work);
	HIPCHK(hipMemsetAsync(c->d_counter, 0, 4, c->stream));
	const uint32_t maxb = k == 0 ? (uint32_t)(c->num_cu * c->bpc_calcd) : s.blocks;
	const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(maxb, (n_work + LANE_BLOCK - 1) / LANE_BLOCK));
	const size_t lds = lane_lds(c);
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	if (c->pos32)
		hipLaunchKernelGGL(kl_calc_d<uint32_t>, dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, dbgD, dbgDs,
		                   c->maxlen + 1, (uint32_t)c->kp.seed_length + 1, c->d_stats);
	else
		hipLaunchKernelGGL(kl_calc_d<uint64_t>, dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, dbgD, dbgDs,
		                   c->maxlen + 1, (uint32_t)c->kp.seed_length + 1, c->d_stats);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	c->stats.ms_calc_d += ms; c->stats.launches_calc_d++;
	if (getenv("BWB_DEBUG")) fprintf(stderr, "[bwb] k_calc_d class %d: %u reads, grid %u, %.3f ms\n", k, n_work, grid, ms);
	return BWB_OK;
}

static int run_calc_d(bwb_hip_ctx *c, int32_t *dbgD, int32_t *dbgDs) {
	int rc = launch_calc_d(c, 0, nullptr, c->n_reads, dbgD, dbgDs);
	if (rc) return rc;
	std::vector<uint32_t> ids;
	for (int k = 1; k <= 2; k++) {
		rc = collect(c, ST_SCRATCH_OVF, ids);
		if (rc) return rc;
		if (ids.empty()) return BWB_OK;
		c->stats.n_overflow_reads += ids.size();
		rc = ensure_class(c, k);
		if (rc) return rc;
		rc = launch_calc_d(c, k, c->d_worklist, (uint32_t)ids.size(), dbgD, dbgDs);
		if (rc) return rc;
	}
	rc = collect(c, ST_SCRATCH_OVF, ids);
	if (rc) return rc;
	if (!ids.empty()) return fail(BWB_E_OVERFLOW, "calculate_d: SA-interval list exceeded the largest scratch class");
	return BWB_OK;
}

static int launch_search(bwb_hip_ctx *c, int k, const uint32_t *wl, uint32_t n_work, uint32_t iter_budget = 0, uint32_t lane_stride = 1) {
	ScratchClass &s = c->cls[k];
	Batch b = make_batch(c, wl, n_work);
	b.iter_budget = iter_budget; b.lane_stride = lane_stride;
	OutBuf ob{ c->d_log, c->d_count, c->log_cap, c->d_off, c->d_n };
	const size_t lds = lane_lds(c);
	HIPCHK(hipMemsetAsync(c->d_counter, 0, 4, c->stream));
	HIPCHK(hipMemsetAsync(c->d_pool_bump, 0, POOL_REGIONS * 64, c->stream)); /* every launch starts with an empty chunk pool */
	const uint32_t per_block = LANE_BLOCK / lane_stride;
	const uint32_t maxb = k == 0 ? (uint32_t)(c->num_cu * c->bpc_search) : s.blocks;
	const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(maxb, (n_work + per_block - 1) / per_block));
	/* one region per 8 blocks up to POOL_REGIONS, so that the few blocks of a small launch (class 2) are not confined to
	 * a fraction of the pool; a state word names 2^26 chunks of its region */
	s.sc.pool = c->d_pool; s.sc.pool_bump = c->d_pool_bump; /* (the pool may have been replaced since the class was set up) */
	s.sc.n_regions = std::max<uint32_t>(1, std::min<uint32_t>(POOL_REGIONS, grid / 8));
	s.sc.region_u4 = c->pool_bytes / s.sc.n_regions / 4096 * 256;
	s.sc.pool_cap = (uint32_t)std::min<size_t>(s.sc.region_u4 * 16 / (c->wide ? 2048 : 1024), (size_t)1 << 26);
	{ /* private runs take at most three quarters of a region */
		const uint32_t lanes = (grid + s.sc.n_regions - 1) / s.sc.n_regions * LANE_BLOCK;
		s.sc.keep = std::min<uint32_t>(c->keep, s.sc.pool_cap / 4 * 3 / lanes);
	}
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	if (c->pos32 && !c->wide)
		hipLaunchKernelGGL((kl_search<uint32_t, false>), dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, ob, c->d_stats);
	else if (c->pos32)
		hipLaunchKernelGGL((kl_search<uint32_t, true>), dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, ob, c->d_stats);
	else if (!c->wide)
		hipLaunchKernelGGL((kl_search<uint64_t, false>), dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, ob, c->d_stats);
	else
		hipLaunchKernelGGL((kl_search<uint64_t, true>), dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, b, c->kp, s.sc, ob, c->d_stats);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	c->stats.ms_search += ms; c->stats.launches_search++;
	if (getenv("BWB_DEBUG")) {
		unsigned int bump[POOL_REGIONS * 16], used_max = 0;
		unsigned long long used = 0;
		hipMemcpy(bump, c->d_pool_bump, sizeof(bump), hipMemcpyDeviceToHost);
		const uint32_t priv = (grid + s.sc.n_regions - 1) / s.sc.n_regions * LANE_BLOCK * s.sc.keep;
		for (uint32_t r = 0; r < s.sc.n_regions; r++) { used += std::min(priv + bump[r * 16], s.sc.pool_cap); used_max = std::max(used_max, priv + bump[r * 16]); }
		fprintf(stderr, "[bwb] k_search class %d: %u reads, grid %u, budget %u, lane stride %u, %.3f ms, pool chunks used %llu of %d x %u (fullest region asked for %u; %u private per lane)\n", k, n_work, grid, iter_budget, lane_stride, ms, used, (int)s.sc.n_regions, s.sc.pool_cap, used_max, s.sc.keep);
	}
	return BWB_OK;
}

/* grows the hit log if reads were refused for lack of room, then re-runs them in class k */
static int rerun_out_overflow(bwb_hip_ctx *c, int k) {
	std::vector<uint32_t> ids;
	for (int guard = 0; guard < 40; guard++) {
		int rc = collect(c, ST_OUT_OVF, ids);
		if (rc) return rc;
		if (ids.empty()) return BWB_OK;
		unsigned long long cnt = 0;
		HIPCHK(hipMemcpy(&cnt, c->d_count, 8, hipMemcpyDeviceToHost));
		const uint64_t valid = std::min<uint64_t>(cnt, c->log_cap);
		const uint64_t ncap = c->log_cap * 4;
		uint4 *nl = nullptr;
		HIPCHK(hipMalloc(&nl, ncap * 32));
		HIPCHK(hipMemcpy(nl, c->d_log, valid * 32, hipMemcpyDeviceToDevice));
		hipFree(c->d_log);
		c->d_log = nl; c->log_cap = ncap;
		cnt = valid;
		HIPCHK(hipMemcpy(c->d_count, &cnt, 8, hipMemcpyHostToDevice));
		rc = launch_search(c, k, c->d_worklist, (uint32_t)ids.size());
		if (rc) return rc;
	}
	return fail(BWB_E_OVERFLOW, "hit log kept overflowing");
}

extern "C" int bwb_hip_batch_run(bwb_hip_ctx *c) {
	if (!c || !c->uploaded) return fail(BWB_E_STATE, "batch_run: no batch uploaded");
	HIPCHK(hipSetDevice(c->device));
	memset(&c->stats, 0, sizeof(c->stats));
	HIPCHK(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * 40, c->stream));
	HIPCHK(hipMemsetAsync(c->d_count, 0, 8, c->stream));
	if (c->n_reads == 0) { c->ran = true; return BWB_OK; }
	HIPCHK(hipMemsetAsync(c->d_n, 0, (size_t)c->n_reads * 4, c->stream));
	hipEvent_t t0, t1;
	HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
	HIPCHK(hipEventRecord(t0, c->stream));
	int rc = run_calc_d(c, nullptr, nullptr);
	if (rc) return rc;
	/* phase 1: every read, with an iteration budget; phase 2: the reads that exceeded it (the heavy tail),
	 * restarted together, one per octet, so that each runs the low-latency cooperative path from the start */
	uint32_t budget = 0; /* off by default: measured slower than one launch (1534 + 770 ms vs 2105 ms, chr21-scale -n 3, 1 M reads) */
	if (getenv("BWB_ITER_BUDGET")) budget = (uint32_t)strtoul(getenv("BWB_ITER_BUDGET"), nullptr, 10);
	rc = launch_search(c, 0, nullptr, c->n_reads, budget, 1);
	if (rc) return rc;
	std::vector<uint32_t> ids;
	if (budget) {
		rc = collect(c, ST_HEAVY, ids);
		if (rc) return rc;
		if (!ids.empty()) {
			c->stats.n_heavy_reads = ids.size();
			rc = launch_search(c, 0, c->d_worklist, (uint32_t)ids.size(), 0, 8);
			if (rc) return rc;
		}
	}
	rc = rerun_out_overflow(c, 0);
	if (rc) return rc;
	for (int k = 1; k <= 2; k++) {
		rc = collect(c, ST_SCRATCH_OVF, ids);
		if (rc) return rc;
		if (ids.empty()) break;
		c->stats.n_overflow_reads += ids.size();
		rc = ensure_class(c, k);
		if (rc) return rc;
		rc = launch_search(c, k, c->d_worklist, (uint32_t)ids.size());
		if (rc) return rc;
		rc = rerun_out_overflow(c, k);
		if (rc) return rc;
	}
	rc = collect(c, ST_SCRATCH_OVF, ids);
	if (rc) return rc;
	if (!ids.empty()) return fail(BWB_E_OVERFLOW, "a read exceeded the largest per-read scratch class");
	HIPCHK(hipEventRecord(t1, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, t0, t1));
	c->stats.ms_total = ms;
	hipEventDestroy(t0); hipEventDestroy(t1);
	unsigned long long st[40];
	HIPCHK(hipMemcpy(st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost));
	c->stats.visits_single = st[STAT_VIS_SINGLE]; c->stats.visits_alphabet = st[STAT_VIS_ALPHA]; c->stats.visits_calc_d = st[STAT_VIS_CALCD];
	c->stats.heap_pops = st[STAT_POPS]; c->stats.heap_pushes = st[STAT_PUSHES]; c->stats.n_alignments = st[STAT_ALNS];
	if (getenv("BWB_DEBUG")) {
		fprintf(stderr, "[bwb] search loop iterations: total %llu, longest lane %llu, wave iterations %llu (%.1f of 64 lanes busy)\n", st[STAT_N], st[STAT_N_MAX], st[16], st[16] ? (double)st[STAT_N] / (double)st[16] : 0.0);
		if (st[24 + 3]) {
			double tot = 0; for (int k = 0; k < 16; k++) tot += (double)st[24 + k];
			const char *nm[16] = { "top", "A(pop)", "B(issue)", "C(rank)", "D.tail", "E(exact)", "-", "F(finish,grab)", "D.prune/hit", "D.masks", "D.reserve", "D.templates", "D.gap", "D.mm/match", "-", "-" };
			fprintf(stderr, "[bwb] stamps (%% of lane cycles):");
			for (int k = 0; k < 16; k++) if (st[24 + k]) fprintf(stderr, " %s %.1f |", nm[k], 100.0 * (double)st[24 + k] / tot);
			fprintf(stderr, "\n");
		}
	}
	c->ran = true;
	return BWB_OK;
}

extern "C" int bwb_hip_get_stats(bwb_hip_ctx *c, bwb_stats *out) {
	if (!c || !out) return fail(BWB_E_ARG, "get_stats: null argument");
	*out = c->stats;
	return BWB_OK;
}

extern "C" int bwb_hip_batch_result(bwb_hip_ctx *c, bwb_result *out) {
	if (!c || !out) return fail(BWB_E_ARG, "batch_result: null argument");
	if (!c->ran) return fail(BWB_E_STATE, "batch_result: batch_run has not completed");
	HIPCHK(hipSetDevice(c->device));
	const uint32_t n = c->n_reads;
	std::vector<uint32_t> cnt(n ? n : 1);
	if (n) HIPCHK(hipMemcpy(cnt.data(), c->d_n, (size_t)n * 4, hipMemcpyDeviceToHost));
	c->h_aln_off.assign((size_t)n + 1, 0);
	for (uint32_t i = 0; i < n; i++) c->h_aln_off[i + 1] = c->h_aln_off[i] + cnt[i];
	const uint64_t total = c->h_aln_off[n];
	c->h_alns.resize(total ? total : 1);
	if (total) {
		if (c->sorted_cap < total) {
			hipFree(c->d_sorted); c->d_sorted = nullptr;
			HIPCHK(hipMalloc(&c->d_sorted, total * 32));
			c->sorted_cap = total;
		}
		HIPCHK(hipMemcpy(c->d_dstoff, c->h_aln_off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_gather, dim3((n + 31) / 32), dim3(256), 0, c->stream, c->d_log, c->d_off, c->d_n, c->d_dstoff, n, c->d_sorted);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(c->stream));
		HIPCHK(hipMemcpy(c->h_alns.data(), c->d_sorted, total * 32, hipMemcpyDeviceToHost));
	}
	out->n_reads = n;
	out->aln_off = c->h_aln_off.data();
	out->alns = c->h_alns.data();
	return BWB_OK;
}

extern "C" int bwb_hip_align_batch(bwb_hip_ctx *c, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                                   uint32_t n_reads, uint32_t stride, bwb_result *out) {
	const bool dbg = getenv("BWB_DEBUG") != nullptr;
	double t0 = wall_s();
	int rc = bwb_hip_batch_upload(c, p, reads_fwd, lens, n_reads, stride);
	if (rc) return rc;
	const double t1 = wall_s();
	rc = bwb_hip_batch_run(c);
	if (rc) return rc;
	const double t2 = wall_s();
	rc = bwb_hip_batch_result(c, out);
	if (dbg) fprintf(stderr, "[bwb] align_batch: upload %.3f s, run %.3f s, result %.3f s\n", t1 - t0, t2 - t1, wall_s() - t2);
	return rc;
}

extern "C" int bwb_hip_calc_d(bwb_hip_ctx *c, int32_t *out_D, int32_t *out_Dseed) {
	if (!c || !out_D || !out_Dseed) return fail(BWB_E_ARG, "calc_d: null argument");
	if (!c->uploaded) return fail(BWB_E_STATE, "calc_d: no batch uploaded");
	HIPCHK(hipSetDevice(c->device));
	const size_t nD = (size_t)c->n_reads * (c->maxlen + 1) * 2, nS = (size_t)c->n_reads * (c->kp.seed_length + 1) * 2;
	int32_t *dD = nullptr, *dS = nullptr;
	HIPCHK(hipMalloc(&dD, (nD ? nD : 1) * 4));
	HIPCHK(hipMalloc(&dS, (nS ? nS : 1) * 4));
	HIPCHK(hipMemset(dD, 0, (nD ? nD : 1) * 4));
	HIPCHK(hipMemset(dS, 0, (nS ? nS : 1) * 4));
	memset(&c->stats, 0, sizeof(c->stats));
	HIPCHK(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * 40, c->stream));
	int rc = c->n_reads ? run_calc_d(c, dD, dS) : BWB_OK;
	if (!rc) {
		hipMemcpy(out_D, dD, nD * 4, hipMemcpyDeviceToHost);
		hipMemcpy(out_Dseed, dS, nS * 4, hipMemcpyDeviceToHost);
	}
	hipFree(dD); hipFree(dS);
	return rc;
}

extern "C" int bwb_hip_rank16(bwb_hip_ctx *c, const uint64_t *pos, size_t n, int inc, int exact, uint64_t *out) {
	if (!c || (n && (!pos || !out))) return fail(BWB_E_ARG, "rank16: null argument");
	if (n == 0) return BWB_OK;
	for (size_t i = 0; i < n; i++)
		if (pos[i] != ~0ull && pos[i] >= c->ix.length) return fail(BWB_E_ARG, "rank16: position out of range");
	HIPCHK(hipSetDevice(c->device));
	uint64_t *dp = nullptr, *dout = nullptr;
	HIPCHK(hipMalloc(&dp, n * 8));
	HIPCHK(hipMalloc(&dout, n * 128));
	HIPCHK(hipMemcpy(dp, pos, n * 8, hipMemcpyHostToDevice));
	const unsigned grid = (unsigned)std::min<size_t>((n + BWB_OCTS_PER_BLOCK - 1) / BWB_OCTS_PER_BLOCK, (size_t)c->num_cu * 8);
	hipLaunchKernelGGL(k_rank16, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, dp, (uint64_t)n, inc, exact, dout);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpy(out, dout, n * 128, hipMemcpyDeviceToHost));
	hipFree(dp); hipFree(dout);
	return BWB_OK;
}

extern "C" int bwb_hip_rank_bench(bwb_hip_ctx *c, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum) {
	if (!c || n == 0 || iters < 1) return fail(BWB_E_ARG, "rank_bench: bad argument");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(hipMemsetAsync(c->d_count, 0, 8, c->stream));
	const unsigned grid = (unsigned)(c->num_cu * 8);
	auto launch = [&](uint64_t sd) {
		if (c->pos32) hipLaunchKernelGGL(k_rank_bench<uint32_t>, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, c->d_count);
		else hipLaunchKernelGGL(k_rank_bench<uint64_t>, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, c->d_count);
	};
	launch(seed); /* warm-up */
	HIPCHK(hipMemsetAsync(c->d_count, 0, 8, c->stream));
	HIPCHK(hipEventRecord(c->ev0, c->stream));
	for (int i = 0; i < iters; i++)
		launch(seed + i);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(c->ev1, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	unsigned long long cs = 0;
	HIPCHK(hipMemcpy(&cs, c->d_count, 8, hipMemcpyDeviceToHost));
	if (ms_per_iter) *ms_per_iter = ms / iters;
	if (checksum) *checksum = cs;
	return BWB_OK;
}

extern "C" int bwb_hip_set_sa(bwb_hip_ctx *c, const uint64_t *SA, uint64_t num_sa) {
	if (!c || !SA || num_sa != (c->ix.length + 31) / 32) return fail(BWB_E_ARG, "set_sa: bad argument");
	HIPCHK(hipSetDevice(c->device));
	hipFree(c->d_SA); c->d_SA = nullptr;
	HIPCHK(hipMalloc(&c->d_SA, num_sa * 8));
	HIPCHK(hipMemcpy(c->d_SA, SA, num_sa * 8, hipMemcpyHostToDevice));
	c->num_sa = num_sa;
	return BWB_OK;
}

extern "C" int bwb_hip_locate(bwb_hip_ctx *c, const uint64_t *rows, size_t n, uint64_t *out_pos) {
	if (!c || (n && (!rows || !out_pos))) return fail(BWB_E_ARG, "locate: null argument");
	if (!c->d_SA) return fail(BWB_E_STATE, "locate: sampled SA not uploaded (bwb_hip_set_sa)");
	if (n == 0) return BWB_OK;
	for (size_t i = 0; i < n; i++) if (rows[i] >= c->ix.length) return fail(BWB_E_ARG, "locate: row out of range");
	HIPCHK(hipSetDevice(c->device));
	uint64_t *dr = nullptr, *dout = nullptr;
	HIPCHK(hipMalloc(&dr, n * 8));
	HIPCHK(hipMalloc(&dout, n * 8));
	HIPCHK(hipMemcpy(dr, rows, n * 8, hipMemcpyHostToDevice));
	const unsigned grid = (unsigned)std::min<size_t>((n + BWB_OCTS_PER_BLOCK - 1) / BWB_OCTS_PER_BLOCK, (size_t)c->num_cu * 8);
	hipLaunchKernelGGL(k_locate, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, c->d_SA, c->sa0_index, dr, (uint64_t)n, dout);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(c->stream));
	HIPCHK(hipMemcpy(out_pos, dout, n * 8, hipMemcpyDeviceToHost));
	hipFree(dr); hipFree(dout);
	return BWB_OK;
}

/* developer aid (not in include/bwbble_hip.h): per-read loop iteration counts of the last batch_run when BWB_DEBUG_ITERS is set */
extern "C" int bwb_hip_debug_iters(bwb_hip_ctx *c, uint32_t *out) {
	if (!c || !c->d_dbg_iters) return fail(BWB_E_STATE, "debug iteration counts are off (set BWB_DEBUG_ITERS)");
	HIPCHK(hipMemcpy(out, c->d_dbg_iters, (size_t)c->n_reads * 4, hipMemcpyDeviceToHost));
	return BWB_OK;
}

/* developer aid: k_calc_d visits per read of the last batch_run */
extern "C" int bwb_hip_debug_calcd_work(bwb_hip_ctx *c, uint32_t *out) {
	if (!c || !c->uploaded) return fail(BWB_E_STATE, "no batch");
	std::vector<uint8_t> h((size_t)c->n_reads * c->dstride);
	HIPCHK(hipMemcpy(h.data(), c->d_dbuf, h.size(), hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < c->n_reads; i++) memcpy(&out[i], &h[(size_t)i * c->dstride + c->dstride - 8], 4);
	return BWB_OK;
}
