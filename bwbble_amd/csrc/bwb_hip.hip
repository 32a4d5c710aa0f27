/*
 * bwb_hip.hip - C-ABI implementation of include/bwbble_hip.h (libbwbble_hip.so), MI355X / gfx950 only.
 *
 * Host orchestration (replaces the per-batch body of align_reads_inexact_parallel, mg-aligner/inexact_match.c:103-165).
 * A context keeps up to BWB_MAX_SLOTS batches of reads resident ("slots") and runs them as a stream:
 *
 *   slot_upload(s)  reads -> pinned staging -> HBM on the copy stream                        (overlaps the kernels)
 *   slot_submit(s)  kl_calc_d over the slot, then ONE SLICE of kl_search fed from the slot's cursor.  With more batches to
 *                   come the slice does not drain: when the cursor runs out, waves park the reads they are working on in
 *                   the lanes' save area (bwb_lane.h) and the next slot's slice resumes them, so the heavy tail of batch k
 *                   runs next to the bulk of batch k+1 instead of on a nearly empty GPU.
 *   slot_wait(s)    until every read of the slot is done (normally when the following slice ends; a draining launch
 *                   otherwise), then the rare reads that did not fit the class-0 per-read scratch are re-run - still on the
 *                   GPU - in class 1, then class 2 (fewer lanes, larger lists, the whole chunk pool after a draining launch)
 *   slot_result(s)  hit log -> host on the result stream, put into read order
 *
 * batch_upload / batch_run / batch_result are the same on slot 0 with a draining slice (one batch, nothing to overlap).
 * There is no CPU fallback anywhere in this file.
 */
#include <hip/hip_runtime.h>
#include <time.h>
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "../../include/bwbble_hip.h"
#include "bwb_kernels.h"
static double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#include "bwb_lane.h"

static thread_local std::string g_err;
static int fail(int code, const std::string &m) { g_err = m; return code; }
#define HIPCHK(x)                                                                                         \
	do {                                                                                                  \
		hipError_t e_ = (x);                                                                              \
		if (e_ != hipSuccess) return fail(BWB_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));  \
	} while (0)

/* device / pinned-host allocations that free themselves: an early error return leaks nothing */
struct DevMem {
	void *p = nullptr;
	size_t bytes = 0;
	DevMem() = default;
	DevMem(const DevMem &) = delete;
	DevMem &operator=(const DevMem &) = delete;
	~DevMem() { release(); }
	void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
	hipError_t alloc(size_t n) { release(); hipError_t e = hipMalloc(&p, n ? n : 1); if (e == hipSuccess) bytes = n; else p = nullptr; return e; }
	/* contents are NOT kept; grows by half again at least, so that a stream of slightly larger batches does not reallocate (hipFree
	 * synchronises the whole device and would stall the queued slices) */
	hipError_t reserve(size_t n) {
		if (p && bytes >= n) return hipSuccess;
		if (p && alloc(std::max(n, bytes + bytes / 2)) == hipSuccess) return hipSuccess;
		return alloc(n); /* (exactly what is asked for, when half again as much does not fit) */
	}
	template <typename T> T *as() const { return (T *)p; }
};
struct PinMem {
	void *p = nullptr;
	size_t bytes = 0;
	PinMem() = default;
	PinMem(const PinMem &) = delete;
	PinMem &operator=(const PinMem &) = delete;
	~PinMem() { if (p) (void)hipHostFree(p); }
	hipError_t reserve(size_t n) {
		if (p && bytes >= n) return hipSuccess;
		if (p) (void)hipHostFree(p);
		p = nullptr; bytes = 0;
		hipError_t e = hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault);
		if (e == hipSuccess) bytes = n; else p = nullptr;
		return e;
	}
	template <typename T> T *as() const { return (T *)p; }
};
struct Event {
	hipEvent_t e = nullptr;
	Event() = default;
	Event(const Event &) = delete;
	Event &operator=(const Event &) = delete;
	~Event() { if (e) (void)hipEventDestroy(e); }
	hipError_t create() { return e ? hipSuccess : hipEventCreate(&e); }
};

/* per-lane scratch of one class (class 0 = every resident lane, classes 1/2 = fewer lanes with larger lists) */
struct ScratchClass {
	DevMem mem;
	LaneScratch sc{};
	uint32_t blocks = 0;
	bool ready = false;
};

/* one resident batch */
struct Slot {
	bool uploaded = false, submitted = false, complete = false, fetched = false;
	uint32_t n_reads = 0, stride = 0, maxlen = 0, dstride = 0;
	DevMem d_reads, d_lens, d_dbuf, d_status, d_n, d_off, d_worklist, d_log, d_ctl, d_dbg_iters, d_src;
	uint32_t n_tot = 0;           /* n_reads, plus one when the carried read (the source of a leading short read's D_seed) rides along for kl_calc_d */
	bool inherit = false;         /* some read takes its D_seed from an earlier one (d_src) */
	std::vector<uint32_t> h_src;
	uint64_t log_cap = 0;
	PinMem h_reads, h_lens, h_cnt, h_off, h_log, h_ctl;
	std::vector<uint8_t> h_status;
	std::vector<uint64_t> h_aln_off;
	std::vector<bwb_aln> h_alns;
	Event ev_up;
	bool calcd_queued = false;    /* kl_calc_d of this batch has been queued ahead of its submit, on the context's second kernel stream (calcd_ahead) */
	Event ev_calcd;
	uint64_t launch = 0;          /* sequence number of the slice that was fed from this slot */
	/* control words on the device (d_ctl, 64 bytes apart): the work cursor, the number of finished reads, the hit count */
	uint32_t *ctl_counter() const { return d_ctl.as<uint32_t>(); }
	unsigned int *ctl_done() const { return d_ctl.as<unsigned int>() + 16; }
	unsigned long long *ctl_count() const { return (unsigned long long *)(d_ctl.as<unsigned char>() + 128); }
	uint32_t *ctl_counter2() const { return d_ctl.as<uint32_t>() + 48; } /* cursor of the re-run launches */
};

struct PendingTime { hipEvent_t e0, e1; int kind; }; /* kind 0 = kl_calc_d, 1 = kl_search */

struct bwb_hip_ctx {
	int device = 0, num_cu = 0;
	hipStream_t stream = nullptr, cstream = nullptr, rstream = nullptr; /* kernels; uploads; results */
	hipStream_t dstream = nullptr;      /* kl_calc_d of the batches AHEAD of the one being searched (calcd_ahead > 0) */
	int calcd_ahead = 0;                /* BWB_CALCD_AHEAD: how many uploaded slots beyond the submitted one get their kl_calc_d queued on dstream at once */
	DevIndex ix{};
	DevMem d_buckets, d_SA, d_stats, d_descs, d_misc;
	uint64_t sa0_index = 0, num_sa = 0;
	bool pos32 = true;                  /* BWT rows fit 32-bit positions */
	bwb_params p{};
	KParams kp{};
	bool have_params = false;
	Slot slots[BWB_MAX_SLOTS];
	SlotDesc h_descs[BWB_MAX_SLOTS]{};
	ScratchClass cls[3];
	DevMem d_pool, d_pool_bump;         /* heap chunk pool, POOL_REGIONS equal regions (the re-run classes use it after a drain); bump counters (two sets) */
	uint32_t keep = 256;                /* chunks of a lane's private run (BWB_KEEP) */
	int bpc_search = LANE_WAVES_PER_SIMD, bpc_calcd = LANE_WAVES_PER_SIMD; /* blocks of four waves per CU = waves per SIMD */
	bool wide = false;                  /* 32-byte heap entries (max_gapo > 1: more than one gap run per path) */
	bool parked = false;                /* reads may be parked in the class-0 save area (the last class-0 launch was a non-draining slice) */
	uint64_t n_launches = 0;            /* class-0 search launches so far */
	std::vector<hipEvent_t> launch_ev;  /* launch_ev[k-1-ev_base]: recorded after class-0 search launch k */
	uint64_t ev_base = 0;               /* launches whose events have been recycled (no slot in flight can wait for them) */
	std::vector<PendingTime> pending;
	std::vector<hipEvent_t> free_events;
	uint32_t slice_iters = 0;           /* BWB_SLICE_ITERS: test knob, time-sliced launches */
	double locate_ms = 0; uint64_t locate_steps = 0, locate_rows = 0; /* the last bwb_hip_locate call */
	bool force_slices = false;          /* BWB_FORCE_SLICES: the one-batch API parks and resumes too (tests) */
	bool dbg = false, dbg_iters = false;
	const char *launch_log = nullptr;   /* BWB_LAUNCH_LOG=<file>: one JSON line per class-0 kernel launch (synchronises after every launch: a profiling aid) */
	unsigned long long log_prev[16] = { 0 };
	bwb_stats stats{};
	double t_run0 = 0;
	/* the calculate_d table (bwb_lane.h: DTab): entries and interval lists of all 4^K K-mers, built at the first large batch */
	DevMem d_dtab_ent, d_dtab_pool;
	int dtab_K = 0;                     /* 0: no table */
	uint32_t dtab_nm1[4] = { 0, 0, 0, 0 }; /* the level-1 entries' summed widths (DTab::nm1) */
	int dtab_multiref = -1;             /* the alphabet it was built for (-S has its own children) */
	int dtab_mode = -1;                 /* BWB_DTAB: 0 never, 1 always, unset: when a batch has at least DTAB_MIN_READS reads */
	bool dtab_failed = false;           /* a build did not fit its buffers: not tried again */
	double dtab_seconds = 0;
	/* bwb_hip_ctx_create_async: the index upload runs on this thread; everything that launches a kernel joins it first (index_ready) */
	std::thread idx_thread;
	int idx_rc = BWB_OK;
	std::string idx_err;
	double idx_seconds = 0, pool_seconds = 0; /* index upload (context creation to the last bucket in HBM); the chunk pool's hipMalloc */
	~bwb_hip_ctx() {
		if (idx_thread.joinable()) idx_thread.join();
		for (auto &pt : pending) { (void)hipEventDestroy(pt.e0); (void)hipEventDestroy(pt.e1); }
		for (auto e : free_events) (void)hipEventDestroy(e);
		for (auto e : launch_ev) (void)hipEventDestroy(e);
		if (stream) (void)hipStreamDestroy(stream);
		if (cstream) (void)hipStreamDestroy(cstream);
		if (rstream) (void)hipStreamDestroy(rstream);
		if (dstream) (void)hipStreamDestroy(dstream);
	}
};

extern "C" const char *bwb_hip_last_error(void) { return g_err.c_str(); }
static_assert(sizeof(bwb_aln) == 16 * ALN_U4, "bwb_aln is ALN_U4 16-byte words (bwb_lane.h)");
extern "C" int bwb_hip_abi_version(void) { return BWB_HIP_ABI_VERSION; }

/* NUMA node of the device's PCIe root (sysfs), -1 when the machine has one node or it cannot be told: `bwbble align` pins the host
 * thread that drives a GPU, and with it the pinned staging buffers that thread allocates, to that node */
extern "C" int bwb_hip_device_numa_node(int device) {
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return -1;
	for (char *q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
	const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
	FILE *f = fopen(path.c_str(), "r");
	if (!f) return -1;
	int node = -1;
	if (fscanf(f, "%d", &node) != 1) node = -1;
	fclose(f);
	return node;
}

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

/* D2H of a few bytes without touching the null stream (which would wait for every queued slice) */
static int fetch(bwb_hip_ctx *c, void *dst, const void *src, size_t n) {
	HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->rstream));
	HIPCHK(hipStreamSynchronize(c->rstream));
	return BWB_OK;
}

/* waits until the caller's loader has brought the first `need` 128-character blocks of the index into the host arrays */
static void wait_blocks(const volatile uint64_t *ready, uint64_t need) {
	if (!ready) return;
	while (__atomic_load_n(ready, __ATOMIC_ACQUIRE) < need) { struct timespec ts = { 0, 200000 }; nanosleep(&ts, nullptr); }
}

extern "C" int bwb_hip_ctx_create(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                                  const uint64_t *O, bwb_hip_ctx **out) {
	return bwb_hip_ctx_create_streamed(device, hdr, C, bwt, O, nullptr, out);
}

/* the upload: re-layout on the GPU, chunk by chunk behind the caller's loader (runs on the creating thread, or on the context's own
 * thread after bwb_hip_ctx_create_async) */
static int index_upload(bwb_hip_ctx *c, uint64_t num_words, uint64_t sa0, const uint64_t *C, const uint32_t *bwt, const uint64_t *O, const volatile uint64_t *blocks_ready) {
	const double t0 = wall_s();
	HIPCHK(hipSetDevice(c->device));
	const uint64_t nblk = c->ix.nblk;
	/* superblock base table (a superblock's row is filled in when the upload below reaches its first block: with a streamed index the
	 * host arrays are still being read) */
	std::vector<uint64_t> sbcount(BWB_NSB_MAX * 16, 0);
	/* 2^20 blocks (128 M characters) per chunk: reference arrays -> 128-character buckets in a staging
	 * buffer (k_relayout) -> the index's 64-character buckets (k_relayout64); the two staging sets alternate so that the copy
	 * of chunk k+1 overlaps the kernels of chunk k */
	const uint64_t CH = 1ull << 20;
	DevMem d_bwt[2], d_O[2], d_b128[2], d_sbc;
	Event ev_k[2];
	for (int t = 0; t < 2; t++) {
		HIPCHK(d_bwt[t].alloc(std::min(CH, nblk) * 64));
		HIPCHK(d_O[t].alloc(std::min(CH, nblk) * 128));
		HIPCHK(d_b128[t].alloc(std::min(CH, nblk) * 128));
		HIPCHK(ev_k[t].create());
	}
	HIPCHK(d_sbc.alloc(sbcount.size() * 8));
	int t = 0;
	for (uint64_t b0 = 0; b0 < nblk; b0 += CH, t ^= 1) {
		const uint64_t nb = std::min(CH, nblk - b0);
		const uint64_t w0 = b0 * 16, nw = std::min(nb * 16, num_words - w0);
		wait_blocks(blocks_ready, b0 + nb); /* streamed: the loader has read this chunk of the .bwt file */
		for (uint64_t blk = b0; blk < b0 + nb; blk += (1ull << BWB_SB_SHIFT) - (blk & ((1ull << BWB_SB_SHIFT) - 1))) {
			if (blk & ((1ull << BWB_SB_SHIFT) - 1)) continue; /* (on to the next superblock start inside this chunk) */
			const uint64_t sb = blk >> BWB_SB_SHIFT;
			const uint32_t first = bwt[blk * 16] >> 28;
			for (int j = 0; j < 16; j++) {
				sbcount[sb * 16 + j] = O[blk * 16 + j] - ((first == (uint32_t)j && !(j == 0 && blk * 128 == sa0)) ? 1 : 0);
				c->ix.base[sb][j] = C[j] + sbcount[sb * 16 + j];
			}
			HIPCHK(hipStreamSynchronize(c->stream)); /* (the kernels queued so far read the table: a superblock is 16 chunks, this is rare) */
			HIPCHK(hipMemcpy(d_sbc.p, sbcount.data(), sbcount.size() * 8, hipMemcpyHostToDevice));
		}
		HIPCHK(hipEventSynchronize(ev_k[t].e)); /* the kernel that read this staging set two chunks ago */
		HIPCHK(hipMemcpyAsync(d_bwt[t].p, bwt + w0, nw * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipMemcpyAsync(d_O[t].p, O + b0 * 16, nb * 128, hipMemcpyHostToDevice, c->stream));
		const uint64_t nthreads = nb * 8;
		hipLaunchKernelGGL(k_relayout, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, d_bwt[t].as<uint32_t>(), d_O[t].as<uint64_t>(), b0, nb, nw,
		                   sa0, d_sbc.as<uint64_t>(), d_b128[t].as<uint4>());
		HIPCHK(hipGetLastError());
		hipLaunchKernelGGL(k_relayout64, dim3((unsigned)((nb * 2 + 255) / 256)), dim3(256), 0, c->stream, d_b128[t].as<uint4>(), nb, c->d_buckets.as<uint4>() + b0 * 16);
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(ev_k[t].e, c->stream));
	}
	HIPCHK(hipStreamSynchronize(c->stream));
	c->idx_seconds = wall_s() - t0;
	return BWB_OK;
}

/* joins the upload thread of an asynchronously created context; its error becomes this call's */
static int index_ready(bwb_hip_ctx *c) {
	if (c->idx_thread.joinable()) c->idx_thread.join();
	if (c->idx_rc) return fail(c->idx_rc, "index upload: " + c->idx_err);
	return BWB_OK;
}

static int ctx_create(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt, const uint64_t *O,
                      const volatile uint64_t *blocks_ready, bool async, bwb_hip_ctx **out) {
	if (!hdr || !C || !bwt || !O || !out) return fail(BWB_E_ARG, "ctx_create: null argument");
	const uint64_t length = hdr[0], num_words = hdr[1], num_occ = hdr[3];
	const uint64_t nblk = (length + 127) / 128;
	if (length < 2 || num_occ != nblk || num_words != (length + 7) / 8) return fail(BWB_E_ARG, "ctx_create: inconsistent .bwt header");
	const uint64_t nsb = (nblk + (1ull << BWB_SB_SHIFT) - 1) >> BWB_SB_SHIFT;
	if (nsb > BWB_NSB_MAX) return fail(BWB_E_ARG, "ctx_create: index larger than the superblock table covers (2^34 characters)");
	HIPCHK(hipSetDevice(device));
	std::unique_ptr<bwb_hip_ctx> c(new bwb_hip_ctx()); /* freed with everything it owns on any early return */
	c->device = device;
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, device));
	c->num_cu = prop.multiProcessorCount;
	HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
	HIPCHK(hipStreamCreateWithFlags(&c->rstream, hipStreamNonBlocking));
	HIPCHK(c->d_buckets.alloc(nblk * 256)); /* 64-character buckets: 2 bytes per BWT character */
	HIPCHK(c->d_stats.alloc(sizeof(unsigned long long) * STAT_WORDS));
	HIPCHK(c->d_descs.alloc(sizeof(SlotDesc) * BWB_MAX_SLOTS));
	HIPCHK(c->d_misc.alloc(256));
	HIPCHK(hipMemsetAsync(c->d_stats.p, 0, sizeof(unsigned long long) * STAT_WORDS, c->stream));
	c->dbg = getenv("BWB_DEBUG") != nullptr;
	c->dbg_iters = getenv("BWB_DEBUG_ITERS") != nullptr;
	if (getenv("BWB_DTAB") && *getenv("BWB_DTAB")) c->dtab_mode = atoi(getenv("BWB_DTAB")) ? 1 : 0;
	c->launch_log = getenv("BWB_LAUNCH_LOG");
	if (c->launch_log && !*c->launch_log) c->launch_log = nullptr;
	if (getenv("BWB_SLICE_ITERS")) c->slice_iters = (uint32_t)strtoul(getenv("BWB_SLICE_ITERS"), nullptr, 10);
	c->force_slices = getenv("BWB_FORCE_SLICES") != nullptr || c->slice_iters != 0;
	if (getenv("BWB_CALCD_AHEAD")) c->calcd_ahead = std::max(0, std::min(BWB_MAX_SLOTS - 1, atoi(getenv("BWB_CALCD_AHEAD"))));
	if (c->calcd_ahead > 0) {
		int lo = 0, hi = 0; /* (numerically lowest = highest priority) */
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
		const char *pe = getenv("BWB_CALCD_PRIO");
		HIPCHK(hipStreamCreateWithPriority(&c->dstream, hipStreamNonBlocking, (pe && atoi(pe) > 0) ? hi : ((pe && atoi(pe) < 0) ? lo : 0)));
	}
	memset(&c->ix, 0, sizeof(c->ix));
	for (uint64_t sb = 0; sb < BWB_NSB_MAX; sb++)
		for (int j = 0; j < 16; j++) c->ix.base[sb][j] = C[j];
	for (int j = 0; j < 16; j++) { c->ix.base[BWB_ROW_NEG][j] = C[j]; c->ix.base[BWB_ROW_END][j] = C[j + 1]; }
	c->ix.buckets = c->d_buckets.as<uint4>();
	c->ix.length = length;
	c->ix.nblk = nblk;
	c->sa0_index = hdr[4];
	c->pos32 = length < 0xFFFFFFFFull && !getenv("BWB_FORCE_POS64");
	if (!async) {
		int rc = index_upload(c.get(), num_words, hdr[4], C, bwt, O, blocks_ready);
		if (rc) return rc;
	} else {
		bwb_hip_ctx *cp = c.get();
		const uint64_t sa0 = hdr[4];
		cp->idx_thread = std::thread([cp, num_words, sa0, C, bwt, O, blocks_ready]() {
			cp->idx_rc = index_upload(cp, num_words, sa0, C, bwt, O, blocks_ready);
			if (cp->idx_rc) cp->idx_err = g_err; /* (g_err is per thread: handed to whoever joins) */
		});
	}
	*out = c.release();
	return BWB_OK;
}

extern "C" int bwb_hip_ctx_create_streamed(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                                           const uint64_t *O, const volatile uint64_t *blocks_ready, bwb_hip_ctx **out) {
	return ctx_create(device, hdr, C, bwt, O, blocks_ready, false, out);
}

extern "C" int bwb_hip_ctx_create_async(int device, const uint64_t hdr[5], const uint64_t C[17], const uint32_t *bwt,
                                        const uint64_t *O, const volatile uint64_t *blocks_ready, bwb_hip_ctx **out) {
	return ctx_create(device, hdr, C, bwt, O, blocks_ready, true, out);
}

extern "C" int bwb_hip_ctx_index_wait(bwb_hip_ctx *c, double *seconds) {
	if (!c) return fail(BWB_E_ARG, "ctx_index_wait: null context");
	int rc = index_ready(c);
	if (rc) return rc;
	if (seconds) *seconds = c->idx_seconds;
	return BWB_OK;
}

extern "C" int bwb_hip_dtab_info(bwb_hip_ctx *c, int *K, double *build_seconds, uint64_t *bytes) {
	if (!c) return fail(BWB_E_ARG, "dtab_info: null context");
	if (K) *K = c->dtab_K;
	if (build_seconds) *build_seconds = c->dtab_seconds;
	if (bytes) *bytes = c->dtab_K ? c->d_dtab_ent.bytes + c->d_dtab_pool.bytes : 0;
	return BWB_OK;
}

extern "C" int bwb_hip_setup_times(bwb_hip_ctx *c, double *index_seconds, double *pool_seconds, uint64_t *pool_bytes) {
	if (!c) return fail(BWB_E_ARG, "setup_times: null context");
	if (index_seconds) *index_seconds = c->idx_thread.joinable() ? -1.0 : c->idx_seconds; /* (-1: the upload is still running) */
	if (pool_seconds) *pool_seconds = c->pool_seconds;
	if (pool_bytes) *pool_bytes = c->d_pool.bytes;
	return BWB_OK;
}

extern "C" void bwb_hip_ctx_destroy(bwb_hip_ctx *c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->idx_thread.joinable()) c->idx_thread.join(); /* (the upload thread reads the caller's arrays: they are free once this returns) */
	(void)hipDeviceSynchronize();
	delete c;
}

static size_t lane_lds(const bwb_hip_ctx *c) {
	(void)c;
	return (size_t)LDS_WAVES_OFF + (size_t)(LANE_BLOCK / 64) * WAVE_LDS_BYTES + LDS_ALIGN_SLACK; /* base table + zero row + per wave: gather staging / children */
}
static size_t calcd_lds(void) { return (size_t)CALCD_LDS_BYTES; } /* kl_calc_d's own map: one base table, its own number of compacted U rows (bwb_lane.h) */

static uint32_t max_reads_resident(const bwb_hip_ctx *c) {
	uint32_t n = 1;
	for (const Slot &s : c->slots) n = std::max(n, s.n_reads);
	return n;
}

/* The class-0 heap chunk pool, sized for the batches at hand and grown when a larger batch or index needs more (it can be
 * replaced only while no read is parked).  hipMalloc costs about 27 ms per GB here, which a 600-read run should not pay. */
static int ensure_pool(bwb_hip_ctx *c) {
	size_t fr = 0, tot = 0;
	HIPCHK(hipMemGetInfo(&fr, &tot));
	fr += c->d_pool.bytes; /* what would be free without the current pool */
	/* Ceiling: what is free (the index, the class-0 scratch and the first slot are allocated by now) minus what is still to
	 * come - the scratch of the re-run classes and their pool, three more slots like the largest so far, some slack - and what
	 * the regions can name: a state word holds a 26-bit chunk index relative to the block's region. */
	const size_t isz = c->pos32 ? 8 : 16;
	const size_t cls1 = (size_t)std::max(1, c->num_cu / 2) * LANE_BLOCK * (2 * 8192 * isz + 1024 * sizeof(bwb_aln)), cls2 = (size_t)LANE_BLOCK * (2 * ((size_t)1 << 20) * isz + 65536 * sizeof(bwb_aln));
	size_t slot_bytes = 0;
	for (const Slot &s : c->slots) slot_bytes = std::max(slot_bytes, s.d_reads.bytes + s.d_dbuf.bytes + s.d_log.bytes + (size_t)s.n_reads * sizeof(bwb_aln));
	/* (every slot still to come like the largest so far: a stream needs them all - the heaviest reads of a batch take several slices'
	 * time, DESIGN.md section 2.3 - and 8 x 3 GB is little next to the pool) */
	size_t more_slots = 0; /* slots that have no buffers yet (a caller that uploads every slot before the first submit - bench.py - has none left) */
	for (const Slot &s : c->slots) if (s.d_reads.bytes == 0) more_slots++;
	/* (the calculate_d table, when it is still to be built: 4^12 entries of 32 bytes and their lists - about sixty intervals each at GRCh37
	 * scale, never more intervals than one and a half times the BWT's rows plus one per K-mer; its BUILD borrows this pool as scratch) */
	const size_t dtab_res = (c->dtab_mode != 0 && !c->dtab_K && !c->dtab_failed)
	                            ? ((size_t)32 << 24) + std::min<size_t>((size_t)64 << 24, (size_t)(c->ix.length + c->ix.length / 2) + ((size_t)1 << 24)) * isz + ((size_t)1 << 28) : 0;
	const size_t reserve = cls1 + cls2 + more_slots * slot_bytes + ((size_t)4 << 30) + dtab_res;
	const size_t ceiling = std::min<size_t>(fr > reserve + ((size_t)1 << 30) ? fr - reserve : fr / 2, (size_t)POOL_REGIONS << 36);
	/* Need: per read in flight, the private run (keep chunks of 1 KB, 2 KB with 32-byte entries) plus a share of the
	 * common part that grows with the index: measured 37 KB per lane at 106 M rows, ~300 KB at 884 M, 870 KB at 6.85 G. */
	const size_t lanes = std::min<size_t>(max_reads_resident(c), (size_t)c->num_cu * (size_t)c->bpc_search * LANE_BLOCK);
	const size_t index_mb = (size_t)(c->ix.nblk >> 13) + 1;
	const size_t per_lane = ((size_t)c->keep << 10) + std::min<size_t>((size_t)1536 << 10, index_mb << 9);
	size_t want = std::max<size_t>((size_t)1 << 30, lanes * per_lane / 4 * 5 * (c->wide ? 2 : 1));
	/* (`-n 0`, the reference's default: a read's heap is its root entry - the search goes straight to the exact tail, :345 - so the pool is a
	 * few chunks per lane instead of 170 GB, whose hipMalloc and hipFree were a quarter of a 10 M-read `align -n 0` run: round 6) */
	if (c->kp.max_diff == 0) want = std::max<size_t>((size_t)256 << 20, lanes * (size_t)4096);
	if (want > ceiling) want = ceiling;
	if (getenv("BWB_POOL_GB") && *getenv("BWB_POOL_GB")) want = std::min<size_t>((size_t)atol(getenv("BWB_POOL_GB")) << 30, fr / 10 * 7);
	if (want < ((size_t)256 << 20)) want = (size_t)256 << 20; /* floor (also what BWB_POOL_GB=0 selects, to test the re-run path: class 2 must still fit a read) */
	want &= ~(size_t)(POOL_REGIONS * 4096 - 1);
	/* (a pool within 15 % of what would be asked for now is kept: `want` moves by a slot's size with every upload and by the table's reserve
	 * once that is built, and giving 170 GB back to the driver and taking them again costs five seconds - rounds 3-5 did that at every one of a
	 * stream's first uploads, round 6's profiles/r6_ab_steps.txt session 3) */
	if (c->d_pool.p && (c->d_pool.bytes >= want || (c->d_pool.bytes >= want / 20 * 17 && !getenv("BWB_POOL_GB")))) return BWB_OK;
	if (c->parked) return BWB_OK; /* parked reads hold chunks of the present pool: keep it (the admission control copes) */
	c->d_pool.release();
	if (want > fr) return fail(BWB_E_HIP, "not enough device memory for the heap chunk pool");
	{ const double t0 = wall_s(); HIPCHK(c->d_pool.alloc(want)); c->pool_seconds += wall_s() - t0; }
	if (c->dbg) fprintf(stderr, "[bwb] chunk pool: %.1f GB allocated in %.2f s\n", (double)want / (1u << 30), c->pool_seconds);
	if (!c->d_pool_bump.p) HIPCHK(c->d_pool_bump.alloc(2 * POOL_REGIONS * 64));
	return BWB_OK;
}

static int ensure_class(bwb_hip_ctx *c, int k) {
	ScratchClass &s = c->cls[k];
	uint32_t blocks, lcap, acap;
	if (k == 0) {
		/* Blocks (of four waves) per CU.  Both kernels are built for three (bwb_lane.h), and run three.  (Rounds 3-5 ran kl_search at two when
		 * the parameters allow more than three differences: every lane holds a read with its heap, and the stationary heaps of 196 608 reads
		 * with -n 5 on 150 bp reads overran the 200 GB chunk pool - reads were abandoned and re-run, 103 k reads/s against 224 k at two blocks,
		 * profiles/r5_bench_line_c5_three_blocks.json.)  The choice can change only while nothing is parked (the grid, and with it the pool
		 * geometry, is fixed for the life of a stream); BWB_BLOCKS_PER_CU overrides it. */
		if (!s.ready || !c->parked) {
			c->bpc_search = LANE_WAVES_PER_SIMD; /* (round 6: also beyond three differences - with the gap entries halved, see bwb_lane.h `combined group`,
			                                        the stationary heaps of 196 608 reads with -n 5 on 150 bp reads fill 60 % of the pool; rounds 3-5 ran two blocks there) */
			c->bpc_calcd = CALCD_WAVES_PER_SIMD;
			if (getenv("BWB_BLOCKS_PER_CU")) c->bpc_search = std::max(1, atoi(getenv("BWB_BLOCKS_PER_CU")));
			if (getenv("BWB_KEEP")) c->keep = (uint32_t)std::max(0, atoi(getenv("BWB_KEEP")));
			if (getenv("BWB_CALCD_BLOCKS_PER_CU")) c->bpc_calcd = std::max(1, atoi(getenv("BWB_CALCD_BLOCKS_PER_CU")));
			/* A slice's blocks must all be resident at once: a parked read only moves while its block runs, and a block that had to
			 * wait for another one to leave would find the cursor exhausted and park again at once.  So never more blocks per CU than
			 * the runtime says fit (registers, LDS) - and never more than the LDS really holds: the runtime's answer was 3 for a block
			 * size of which the hardware placed 2 (allocation in units of 1 280 bytes; profiles/r4_lds_probe.txt), which cost 13 % unnoticed. */
			int occ = 0;
			auto lds_fit = [&](const void *f, size_t dyn) {
				hipFuncAttributes fa;
				if (hipFuncGetAttributes(&fa, f) != hipSuccess) return LANE_WAVES_PER_SIMD;
				const size_t need = ((fa.sharedSizeBytes + dyn + LDS_GRANULE - 1) / LDS_GRANULE) * LDS_GRANULE;
				return (int)std::max<size_t>(1, LDS_CU_BYTES / need);
			};
			const void *kf = c->pos32 ? (c->wide ? (const void *)kl_search<uint32_t, true, true> : (const void *)kl_search<uint32_t, false, true>)
			                          : (c->wide ? (const void *)kl_search<uint64_t, true, true> : (const void *)kl_search<uint64_t, false, true>); /* (the -S instantiations need no more) */
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, LANE_BLOCK, lane_lds(c)) == hipSuccess && occ >= 1) {
				if (c->dbg && !s.ready) fprintf(stderr, "[bwb] kl_search: %d block(s) of %d threads fit a CU\n", occ, LANE_BLOCK);
				c->bpc_search = std::min(c->bpc_search, std::min(occ, lds_fit(kf, lane_lds(c))));
				if (c->dbg && !s.ready && lds_fit(kf, lane_lds(c)) < occ) fprintf(stderr, "[bwb] kl_search: only %d block(s) per CU by LDS granules\n", lds_fit(kf, lane_lds(c)));
			}
			const void *kd = c->pos32 ? (const void *)kl_calc_d<uint32_t> : (const void *)kl_calc_d<uint64_t>;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kd, LANE_BLOCK, calcd_lds()) == hipSuccess && occ >= 1) {
				c->bpc_calcd = std::min(c->bpc_calcd, std::min(occ, lds_fit(kd, calcd_lds())));
				if (c->dbg && !s.ready) fprintf(stderr, "[bwb] kl_calc_d: %d block(s) per CU (occupancy query %d, LDS granules %d)\n", c->bpc_calcd, occ, lds_fit(kd, calcd_lds()));
			}
		}
		/* (the scratch is sized for the larger grid: the two kernels share it) */
		blocks = (uint32_t)(c->num_cu * std::max(LANE_WAVES_PER_SIMD, std::max(c->bpc_search, c->bpc_calcd))); lcap = 4096; acap = 256;
	} else if (k == 1) {
		blocks = (uint32_t)std::max(1, c->num_cu / 2); lcap = 8192; acap = 1024;
	} else {
		blocks = 1; lcap = 1u << 20; acap = 1u << 16;
	}
	const uint32_t nslots = blocks * LANE_BLOCK;
	const size_t isz = c->pos32 ? 8 : 16;
	auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
	const uint32_t brow = std::max<uint32_t>(BSTATE_ROW_MIN, ((uint32_t)c->kp.num_buckets + 63u) & ~63u);
	const size_t b_bstate = al((size_t)brow * nslots * 4), b_lists = al((size_t)nslots * 2 * lcap * isz), b_alns = al((size_t)nslots * acap * sizeof(bwb_aln)),
	             b_save = al((size_t)nslots * SAVE_U4 * 16), b_bsave = al((size_t)blocks * 16);
	/* class 0 runs kl_calc_d of the next batch while parked reads keep their lists: it gets a second set; the re-run classes
	 * are drained before anything else uses them */
	const size_t bytes = b_bstate + b_lists * (k == 0 ? 2 : 1) + b_alns + b_save + b_bsave;
	if (!(s.mem.p && s.mem.bytes >= bytes)) {
		if (k == 0 && c->parked) return fail(BWB_E_STATE, "the class-0 scratch cannot grow while reads are parked (flush first)");
		if (c->dstream) HIPCHK(hipStreamSynchronize(c->dstream)); /* (a kl_calc_d queued ahead works in this scratch) */
		s.mem.release();
		size_t fr = 0, tot = 0;
		HIPCHK(hipMemGetInfo(&fr, &tot));
		if (bytes + ((size_t)1 << 30) > fr) return fail(BWB_E_HIP, "not enough device memory for the per-lane scratch (class " + std::to_string(k) + ")");
		HIPCHK(s.mem.alloc(bytes));
		HIPCHK(hipMemsetAsync((unsigned char *)s.mem.p + bytes - b_save - b_bsave, 0, b_save + b_bsave, c->stream));
	}
	unsigned char *base = s.mem.as<unsigned char>();
	s.sc.bstate = (uint32_t *)base; base += b_bstate;
	s.sc.lists = (void *)base; base += b_lists;
	s.sc.lists_d = s.sc.lists;
	if (k == 0) { s.sc.lists_d = (void *)base; base += b_lists; }
	s.sc.alns = (uint4 *)base; base += b_alns;
	s.sc.save = (uint4 *)base; base += b_save;
	s.sc.blocksave = (uint32_t *)base;
	s.sc.nslots = nslots; s.sc.lcap = lcap; s.sc.acap = acap; s.sc.brow = brow;
	s.sc.keep = c->keep;
	s.blocks = blocks;
	s.ready = true;
	return ensure_pool(c); /* (after the scratch: the pool takes what is left over; the re-run classes use it too, after a drain) */
}

static hipEvent_t get_event(bwb_hip_ctx *c) {
	if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;
	return e;
}

/* adds the HIP-event time of every finished launch to the statistics (all of them when `all`: the stream must be idle then) */
static int resolve_times(bwb_hip_ctx *c, bool all) {
	size_t w = 0;
	for (size_t i = 0; i < c->pending.size(); i++) {
		PendingTime &pt = c->pending[i];
		const bool ready = all || hipEventQuery(pt.e1) == hipSuccess;
		if (!ready) { c->pending[w++] = pt; continue; }
		if (all) HIPCHK(hipEventSynchronize(pt.e1));
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, pt.e0, pt.e1));
		if (pt.kind == 0) { c->stats.ms_calc_d += ms; c->stats.launches_calc_d++; }
		else { c->stats.ms_search += ms; c->stats.launches_search++; }
		c->free_events.push_back(pt.e0); c->free_events.push_back(pt.e1);
	}
	c->pending.resize(w);
	return BWB_OK;
}

static int check_params(const bwb_params *p, int *nb_out) {
	if (p->max_gapo < 0 || p->max_gapo > BWB_MAX_GAP_RUNS) return fail(BWB_E_ARG, "max_gapo (-o) must be in [0,8] on the GPU path (a heap entry and a hit record hold eight gap runs)");
	if (p->max_gape < 0 || p->max_gape > 100 || p->max_diff < 0 || p->max_diff > 100) return fail(BWB_E_ARG, "max_gape/max_diff out of the supported range [0,100]");
	if (p->seed_length < 0 || p->seed_length > 255) return fail(BWB_E_ARG, "seed_length must be in [0,255]");
	if (p->mm_score < 0 || p->gapo_score < 0 || p->gape_score < 0) return fail(BWB_E_ARG, "negative penalties are not supported");
	if (p->mm_score > 255 || p->gapo_score > 255 || p->gape_score > 255) return fail(BWB_E_ARG, "penalties (-M, -O, -E) above 255 are not supported on the GPU path");
	const int nb = (p->max_diff + 1) * p->mm_score + (p->max_gapo + 1) * p->gapo_score + (p->max_gape + 1) * p->gape_score; /* heap_init :513 */
	if (nb < 1 || nb > 1024) return fail(BWB_E_ARG, "score range (heap buckets = (n+1) M + (o+1) O + (e+1) E) must be in [1,1024]");
	if (p->max_entries < 1) return fail(BWB_E_ARG, "max_entries must be positive");
	*nb_out = nb;
	return BWB_OK;
}

static bool any_in_flight(const bwb_hip_ctx *c) {
	for (const Slot &s : c->slots) if (s.submitted && !s.complete) return true;
	return false;
}

static int slot_wait(bwb_hip_ctx *c, int si);

extern "C" int bwb_hip_flush(bwb_hip_ctx *c) {
	if (!c) return fail(BWB_E_ARG, "flush: null context");
	HIPCHK(hipSetDevice(c->device));
	for (int s = 0; s < BWB_MAX_SLOTS; s++)
		if (c->slots[s].submitted && !c->slots[s].complete) { int rc = slot_wait(c, s); if (rc) return rc; }
	HIPCHK(hipStreamSynchronize(c->stream));
	return resolve_times(c, true);
}

extern "C" int bwb_hip_slot_upload(bwb_hip_ctx *c, int si, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                                   uint32_t n_reads, uint32_t stride, const uint8_t *carry_seq, uint32_t carry_len) {
	if (!c || !p || si < 0 || si >= BWB_MAX_SLOTS || (n_reads && (!reads_fwd || !lens)) || stride == 0) return fail(BWB_E_ARG, "slot_upload: bad argument");
	int nb = 0;
	int rc = check_params(p, &nb);
	if (rc) return rc;
	HIPCHK(hipSetDevice(c->device));
	Slot &s = c->slots[si];
	if (s.submitted && !s.complete) { rc = slot_wait(c, si); if (rc) return rc; } /* the slot is being reused */
	if (s.uploaded) HIPCHK(hipEventSynchronize(s.ev_up.e)); /* its previous H2D copy has left the pinned staging buffers */
	if (s.calcd_queued && !s.submitted) HIPCHK(hipStreamSynchronize(c->dstream)); /* (uploaded, its kl_calc_d queued ahead, never submitted: that kernel reads the buffers) */
	s.calcd_queued = false;
	/* every slot in flight runs under the same parameters (they are launch arguments) */
	if (c->have_params && memcmp(&c->p, p, sizeof(*p)) != 0 && (any_in_flight(c) || c->parked)) { rc = bwb_hip_flush(c); if (rc) return rc; }
	/* 32-byte heap entries: more than one gap run per path - and penalties above 63, whose buckets can lie beyond the 64-bucket window of
	 * the non-empty buckets: only these kernels carry the code for that (LHeap::far) */
	const bool wide = p->max_gapo > 1 || p->mm_score > 63 || p->gapo_score > 63 || p->gape_score > 63;
	if (c->cls[0].ready && wide != c->wide && c->parked) { rc = bwb_hip_flush(c); if (rc) return rc; }
	c->p = *p; c->have_params = true;
	c->kp = KParams{ p->max_diff, p->max_gapo, p->max_gape, p->max_entries, p->mm_score, p->gapo_score, p->gape_score,
	                 p->seed_length, p->max_diff_seed, p->max_best, p->no_indel_length, nb, p->use_precalc ? 1 : 0, p->is_multiref ? 1 : 0 };
	c->wide = wide;

	/* Reads the kernels cannot represent (longer than 255 bases: aln_entry_t.i is 8-bit, align.h:104; shorter than 12 with -P:
	 * read2index, align.c:174-186, reads before the buffer) get an empty record and a warning instead of failing the batch. */
	uint32_t maxlen = 0, n_bad = 0;
	/* D_seed of the reads that are not longer than the seed: that of the last longer read before them (k_dseed_inherit).
	 * `carry` = that read for the head of the batch, when the caller knows one from earlier in the file. */
	const uint32_t sl = (uint32_t)p->seed_length;
	auto is_source = [&](const uint8_t *seq, uint32_t len) { /* computes D_seed (inexact_match.c:62-64) and is not skipped by -P (:50-57) */
		if (!(sl && len > sl) || len > 255) return false;
		if (p->use_precalc) { if (len < PRECALC_LEN) return false; for (int k = 0; k < PRECALC_LEN; k++) if (seq[k] > 3) return false; }
		return true;
	};
	const bool ghost = carry_seq && is_source(carry_seq, carry_len) && n_reads;
	uint32_t last = ghost ? n_reads : NONE32;
	s.inherit = false;
	s.h_src.clear();
	for (uint32_t i = 0; i < n_reads; i++) {
		const bool bad = lens[i] > 255 || lens[i] > stride || (p->use_precalc && lens[i] < PRECALC_LEN);
		if (bad) { n_bad++; continue; }
		maxlen = std::max<uint32_t>(maxlen, lens[i]);
		if (is_source(reads_fwd + (size_t)i * stride, lens[i])) last = i;
		else if (sl && lens[i] <= sl && last != NONE32) {
			if (!s.inherit) { s.h_src.assign(n_reads, NONE32); s.inherit = true; }
			s.h_src[i] = last;
		}
	}
	const bool use_ghost = ghost && s.inherit; /* (the carried read only matters when a short read precedes the batch's first longer one) */
	if (s.inherit && !use_ghost) for (uint32_t i = 0; i < n_reads; i++) if (s.h_src[i] == n_reads) s.h_src[i] = NONE32;
	if (s.inherit) { bool any = false; for (uint32_t v : s.h_src) any |= v != NONE32; s.inherit = any; }
	if (use_ghost) maxlen = std::max<uint32_t>(maxlen, carry_len);
	s.n_tot = n_reads + (use_ghost && s.inherit ? 1u : 0u);
	if (n_bad) fprintf(stderr, "[bwbble_hip] warning: %u read(s) longer than 255 bases%s get an empty alignment record\n", n_bad, p->use_precalc ? " or shorter than 12 (-P)" : "");
	s.n_reads = n_reads; s.stride = std::min<uint32_t>(stride, std::max<uint32_t>(maxlen, 1)); s.maxlen = maxlen;
	s.dstride = REC_BYTES * rec_count(maxlen) + 16; /* per read: its records (bwb_kernels.h: rec_put), then 16 bytes (work, N count) */
	const size_t nr = s.n_tot ? s.n_tot : 1;
	HIPCHK(s.d_reads.reserve(nr * s.stride));
	HIPCHK(s.d_lens.reserve(nr * 2));
	HIPCHK(s.d_dbuf.reserve(nr * s.dstride));
	HIPCHK(s.d_status.reserve(nr));
	HIPCHK(s.d_worklist.reserve(nr * 4));
	HIPCHK(s.d_n.reserve(nr * 4));
	HIPCHK(s.d_off.reserve(nr * 8));
	HIPCHK(s.d_ctl.reserve(256));
	HIPCHK(s.h_ctl.reserve(256));
	HIPCHK(s.ev_up.create());
	const uint64_t want = std::max<uint64_t>((uint64_t)nr * 8, 1u << 16); /* hit log: 8 records per read, grown on demand */
	if (s.log_cap < want) { HIPCHK(s.d_log.alloc(want * sizeof(bwb_aln))); s.log_cap = want; }
	if (c->dbg_iters) { HIPCHK(s.d_dbg_iters.reserve(nr * 4)); HIPCHK(hipMemsetAsync(s.d_dbg_iters.p, 0, nr * 4, c->cstream)); }
	/* staging: the caller's buffers are free again when this returns; the copy to HBM proceeds on the copy stream */
	HIPCHK(s.h_reads.reserve(nr * s.stride));
	HIPCHK(s.h_lens.reserve(nr * 2));
	uint8_t *hr = s.h_reads.as<uint8_t>();
	uint16_t *hl = s.h_lens.as<uint16_t>();
	if (s.stride == stride && !n_bad) { memcpy(hr, reads_fwd, (size_t)n_reads * stride); memcpy(hl, lens, (size_t)n_reads * 2); }
	else
		for (uint32_t i = 0; i < n_reads; i++) {
			const bool bad = lens[i] > 255 || lens[i] > stride || (p->use_precalc && lens[i] < PRECALC_LEN);
			hl[i] = bad ? (uint16_t)BAD_LEN : lens[i];
			memcpy(hr + (size_t)i * s.stride, reads_fwd + (size_t)i * stride, bad ? 0 : lens[i]);
		}
	if (s.n_tot > n_reads) { hl[n_reads] = (uint16_t)carry_len; memcpy(hr + (size_t)n_reads * s.stride, carry_seq, carry_len); }
	if (s.n_tot) {
		HIPCHK(hipMemcpyAsync(s.d_reads.p, hr, (size_t)s.n_tot * s.stride, hipMemcpyHostToDevice, c->cstream));
		HIPCHK(hipMemcpyAsync(s.d_lens.p, hl, (size_t)s.n_tot * 2, hipMemcpyHostToDevice, c->cstream));
	}
	if (s.inherit) {
		HIPCHK(s.d_src.reserve((size_t)n_reads * 4));
		HIPCHK(hipMemcpyAsync(s.d_src.p, s.h_src.data(), (size_t)n_reads * 4, hipMemcpyHostToDevice, c->cstream));
		HIPCHK(hipStreamSynchronize(c->cstream)); /* (h_src is pageable) */
	}
	/* the slot's entry of the device table */
	SlotDesc &d = c->h_descs[si];
	d.b.reads = s.d_reads.as<uint8_t>(); d.b.lens = s.d_lens.as<uint16_t>(); d.b.n_reads = n_reads; d.b.stride = s.stride;
	d.b.dbuf = s.d_dbuf.as<uint8_t>(); d.b.dstride = s.dstride; d.b.status = s.d_status.as<uint8_t>(); d.b.dbg_iters = s.d_dbg_iters.as<uint32_t>();
	d.out.alns = s.d_log.as<uint4>(); d.out.count = s.ctl_count(); d.out.cap = s.log_cap; d.out.off = s.d_off.as<uint64_t>(); d.out.n = s.d_n.as<uint32_t>();
	d.done = s.ctl_done();
	/* (pageable source: the runtime stages it before returning, so h_descs may change again right away) */
	HIPCHK(hipMemcpyAsync(c->d_descs.as<SlotDesc>() + si, &d, sizeof(SlotDesc), hipMemcpyHostToDevice, c->cstream));
	HIPCHK(hipEventRecord(s.ev_up.e, c->cstream));
	s.uploaded = true; s.submitted = false; s.complete = false; s.fetched = false;
	return ensure_class(c, 0);
}

/* the table as the kernels see it (none: the table of another alphabet, or not built) */
template <typename P> static DTab<P> dtab_of(const bwb_hip_ctx *c) {
	DTab<P> t;
	const bool use = c->dtab_K > 0 && c->dtab_multiref == (c->kp.multiref ? 1 : 0);
	t.ent = use ? c->d_dtab_ent.as<uint4>() : nullptr;
	t.pool = c->d_dtab_pool.as<Intv<P>>();
	t.K = c->dtab_K;
	for (int j = 0; j < 4; j++) t.nm1[j] = c->dtab_nm1[j];
	return t;
}

/* Builds the calculate_d table (bwb_lane.h: DTab, k_dtab_level) for the current alphabet: K levels, each one step from the one before.  The
 * levels' interval lists ping-pong between the two halves of the heap chunk pool, which is idle - nothing is parked, no slot in flight -
 * when this runs (the first large batch of a context); the last level is copied into an allocation of its own size.  A build that does not
 * fit (pool halves, memory) leaves the context without a table: kl_calc_d then computes every step, as before. */
#define DTAB_MIN_READS 200000u
static bool any_in_flight(const bwb_hip_ctx *c);
template <typename P> static int build_dtab_t(bwb_hip_ctx *c, int K) {
	const double t0 = wall_s();
	ScratchClass &sc = c->cls[0];
	const size_t isz = sizeof(Intv<P>);
	/* scratch for the levels' lists: the two halves of the (idle) chunk pool when they are large enough for any index of this size - never more
	 * intervals than one and a half times the BWT's rows plus one per K-mer, 2^30 at most - else (`-n 0` runs keep a small pool) a temporary
	 * allocation of that size; when that fails, whatever the pool offers (a build that outgrows it leaves no table) */
	const size_t need_half = std::min<size_t>((size_t)64 << 24, (size_t)(c->ix.length + c->ix.length / 2) + ((size_t)1 << (2 * K))) * isz;
	DevMem tmp;
	unsigned char *pool0 = c->d_pool.as<unsigned char>();
	size_t half = (c->d_pool.bytes / 2) & ~(size_t)255;
	if (half < need_half) {
		if (tmp.alloc(2 * need_half) == hipSuccess) { pool0 = tmp.as<unsigned char>(); half = need_half; }
		else (void)hipGetLastError();
	}
	const unsigned long long cap = half / isz;
	DevMem entA, entB, bump;
	const size_t nK = (size_t)1 << (2 * K);
	HIPCHK(entA.alloc(nK * 32));
	HIPCHK(entB.alloc(std::max<size_t>(nK / 4, 1) * 32));
	HIPCHK(bump.alloc(8));
	unsigned char *pool1 = pool0 + half;
	/* level 0: the empty suffix - one interval, the whole index; no restart, no visit */
	const Intv<P> root{ (P)0, (P)(c->ix.length - 1) };
	const uint4 e0[2] = { make_uint4(0u, (1u << 15) | (1u << 16), 0u, 0u), make_uint4(0u, 0u, 0u, 0u) };
	/* levels alternate between the entry arrays so that level K lands in entA, and between the pool halves */
	DevMem *ent_of[2] = { (K & 1) ? &entB : &entA, (K & 1) ? &entA : &entB }; /* ent_of[k & 1] */
	unsigned char *pool_of[2] = { pool0, pool1 };
	HIPCHK(hipMemcpyAsync(ent_of[0]->p, e0, sizeof(e0), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipMemcpyAsync(pool_of[0], &root, sizeof(root), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	const uint32_t grid = (uint32_t)(c->num_cu * c->bpc_calcd);
	unsigned long long total = 0;
	for (int k = 1; k <= K; k++) {
		HIPCHK(hipMemsetAsync(bump.p, 0, 8, c->stream));
		const uint64_t tasks = 1ull << (2 * k);
		const uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid, (tasks + LANE_BLOCK - 1) / LANE_BLOCK));
		hipLaunchKernelGGL(k_dtab_level<P>, dim3(g), dim3(LANE_BLOCK), calcd_lds(), c->stream, c->ix, c->kp.multiref, k, ent_of[(k - 1) & 1]->template as<uint4>(),
		                   (const Intv<P> *)pool_of[(k - 1) & 1], ent_of[k & 1]->template as<uint4>(), (Intv<P> *)pool_of[k & 1], bump.as<unsigned long long>(), cap, sc.sc);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(c->stream)); /* (the level is complete: its interval count) */
		int rc = fetch(c, &total, bump.p, 8);
		if (rc) return rc;
		if (k == 1) { /* the first step's summed width by base: what a lookup after a restart compares the restart's width with */
			uint4 e1[8];
			rc = fetch(c, e1, ent_of[1]->p, sizeof(e1));
			if (rc) return rc;
			for (int j = 0; j < 4; j++) c->dtab_nm1[j] = e1[2 * j].z;
		}
		if (total > cap) {
			if (c->dbg) fprintf(stderr, "[bwb] calculate_d table: level %d needs %llu intervals, the scratch holds %llu: no table\n", k, total, cap);
			c->dtab_failed = true;
			return BWB_OK;
		}
	}
	/* the last level's lists into their own allocation (+ 64 bytes: kl_calc_d reads a list in groups of four intervals) */
	if (c->d_dtab_pool.alloc((size_t)total * isz + 64) != hipSuccess) { (void)hipGetLastError(); c->dtab_failed = true; return BWB_OK; }
	HIPCHK(hipMemcpyAsync(c->d_dtab_pool.p, pool_of[K & 1], (size_t)total * isz, hipMemcpyDeviceToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->d_dtab_ent.release();
	std::swap(c->d_dtab_ent.p, entA.p); std::swap(c->d_dtab_ent.bytes, entA.bytes);
	c->dtab_K = K; c->dtab_multiref = c->kp.multiref ? 1 : 0;
	c->dtab_seconds = wall_s() - t0;
	if (c->dbg) fprintf(stderr, "[bwb] calculate_d table: K = %d, %llu intervals (%.2f GB + %.2f GB of entries), built in %.3f s\n", K, total, (double)total * isz / (1u << 30),
	                    (double)nK * 32 / (1u << 30), c->dtab_seconds);
	return BWB_OK;
}
static int ensure_dtab(bwb_hip_ctx *c, uint32_t n_reads) {
	if (c->dtab_mode == 0 || c->dtab_failed) return BWB_OK;
	if (c->dtab_K && c->dtab_multiref == (c->kp.multiref ? 1 : 0)) return BWB_OK;
	if (c->dtab_mode < 0 && n_reads < DTAB_MIN_READS) return BWB_OK;
	if (c->parked || any_in_flight(c) || !c->d_pool.p || !c->cls[0].ready) return BWB_OK; /* (the pool and the list scratch must be idle: next chance at the next idle submit) */
	if (c->dstream) HIPCHK(hipStreamSynchronize(c->dstream));
	HIPCHK(hipStreamSynchronize(c->stream));
	int K = DTAB_KMAX;
	if (getenv("BWB_DTAB_K")) K = std::max(1, std::min(DTAB_KMAX, atoi(getenv("BWB_DTAB_K"))));
	if (c->dtab_K) { c->d_dtab_ent.release(); c->d_dtab_pool.release(); c->dtab_K = 0; } /* (another alphabet) */
	return c->pos32 ? build_dtab_t<uint32_t>(c, K) : build_dtab_t<uint64_t>(c, K);
}

/* BWB_LAUNCH_LOG: what ONE launch did - its HIP-event time and the counters it added (buckets fetched, heap entries stored / loaded, records
 * loaded) - so that per-dispatch PMC counters (tools/pmc_traffic.sh) can be priced launch by launch, slices and the draining launch apart */
static int log_launch(bwb_hip_ctx *c, const char *kernel, int k, int si, bool drains, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
	HIPCHK(hipStreamSynchronize(st));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	unsigned long long cur[16];
	int rc = fetch(c, cur, c->d_stats.p, sizeof(cur));
	if (rc) return rc;
	if (FILE *f = fopen(c->launch_log, "a")) {
		auto d = [&](int i) { return cur[i] - c->log_prev[i]; };
		fprintf(f, "{\"kernel\": \"%s\", \"class\": %d, \"slot\": %d, \"drains\": %s, \"ms\": %.3f, \"buckets\": %llu, \"entries_stored\": %llu, \"entries_loaded\": %llu, \"records_loaded\": %llu}\n",
		        kernel, k, si, drains ? "true" : "false", ms, d(kernel[3] == 'c' ? STAT_BKT_CALCD : STAT_BKT_SEARCH), d(STAT_ENT_ST), d(STAT_ENT_LD), d(STAT_REC_LD));
		fclose(f);
	}
	memcpy(c->log_prev, cur, sizeof(cur));
	return BWB_OK;
}

static int launch_calc_d(bwb_hip_ctx *c, int k, int si, const uint32_t *wl, uint32_t n_work, uint32_t *counter, int32_t *dbgD, int32_t *dbgDs, hipStream_t st = nullptr) {
	ScratchClass &sc = c->cls[k];
	Slot &s = c->slots[si];
	if (!st) st = c->stream;
	Work wk{ wl, n_work, counter, (uint32_t)si, 0, 0, 0 };
	HIPCHK(hipMemsetAsync(counter, 0, 4, st));
	const uint32_t maxb = k == 0 ? (uint32_t)(c->num_cu * c->bpc_calcd) : sc.blocks;
	const uint32_t grid = std::max<uint32_t>(1, std::min<uint32_t>(maxb, (n_work + LANE_BLOCK - 1) / LANE_BLOCK));
	const size_t lds = calcd_lds();
	hipEvent_t e0 = get_event(c), e1 = get_event(c);
	if (!e0 || !e1) return fail(BWB_E_HIP, "hipEventCreate failed");
	c->pending.push_back(PendingTime{ e0, e1, 0 });
	HIPCHK(hipEventRecord(e0, st));
	if (c->pos32)
		hipLaunchKernelGGL(kl_calc_d<uint32_t>, dim3(grid), dim3(LANE_BLOCK), lds, st, c->ix, c->h_descs[si].b, wk, c->kp, sc.sc, dbgD, dbgDs,
		                   s.maxlen + 1, (uint32_t)c->kp.seed_length + 1, c->d_stats.as<unsigned long long>(), dtab_of<uint32_t>(c));
	else
		hipLaunchKernelGGL(kl_calc_d<uint64_t>, dim3(grid), dim3(LANE_BLOCK), lds, st, c->ix, c->h_descs[si].b, wk, c->kp, sc.sc, dbgD, dbgDs,
		                   s.maxlen + 1, (uint32_t)c->kp.seed_length + 1, c->d_stats.as<unsigned long long>(), dtab_of<uint64_t>(c));
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(e1, st));
	if (c->dbg) {
		HIPCHK(hipStreamSynchronize(st));
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, e0, e1));
		fprintf(stderr, "[bwb] kl_calc_d class %d slot %d: %u reads, grid %u, %.3f ms\n", k, si, n_work, grid, ms);
	}
	if (c->launch_log) return log_launch(c, "kl_calc_d", k, si, false, st, e0, e1);
	return BWB_OK;
}

/* reads not longer than the seed take the D_seed bounds of the last longer read before them (k_dseed_inherit) */
static int launch_inherit(bwb_hip_ctx *c, int si, hipStream_t st = nullptr) {
	Slot &s = c->slots[si];
	if (!s.inherit) return BWB_OK;
	hipLaunchKernelGGL(k_dseed_inherit, dim3((s.n_reads + 255) / 256), dim3(256), 0, st ? st : c->stream, c->h_descs[si].b, s.d_src.as<uint32_t>(), s.n_reads);
	HIPCHK(hipGetLastError());
	return BWB_OK;
}

/* one launch of kl_search in class k: new reads come from `wl`/`n_work` of slot si (n_work 0: only parked reads progress) */
static int launch_search(bwb_hip_ctx *c, int k, int si, const uint32_t *wl, uint32_t n_work, uint32_t *counter, bool suspend, bool reset_counter = true,
                         bool force_drain = false) {
	ScratchClass &s = c->cls[k];
	const bool resume = k == 0 && c->parked;
	if (k != 0 && c->parked) return fail(BWB_E_STATE, "a re-run class was launched while reads are parked");
	const uint32_t slice_iters = (k == 0 && !force_drain) ? c->slice_iters : 0; /* (force_drain: also the time-sliced test mode runs to the end) */
	Work wk{ wl, n_work, counter, (uint32_t)si, suspend ? 1u : 0u, slice_iters, resume ? 1u : 0u };
	const size_t lds = lane_lds(c);
	unsigned int *bump = c->d_pool_bump.as<unsigned int>() + (k == 0 ? 0 : POOL_REGIONS * 16);
	if (reset_counter) HIPCHK(hipMemsetAsync(counter, 0, 4, c->stream));
	if (!resume) HIPCHK(hipMemsetAsync(bump, 0, POOL_REGIONS * 64, c->stream)); /* nothing is parked: the launch starts with an empty chunk pool */
	/* class 0 always runs its full grid: every lane's save word is rewritten by every launch, and the pool geometry
	 * (regions, private runs) must not change while reads are parked */
	const uint32_t grid = k == 0 ? (uint32_t)(c->num_cu * c->bpc_search) : std::max<uint32_t>(1, std::min<uint32_t>(s.blocks, (n_work + LANE_BLOCK - 1) / LANE_BLOCK));
	DevMem &pool = c->d_pool;
	s.sc.pool = pool.as<uint4>(); s.sc.pool_bump = bump;
	/* one region per 8 blocks up to POOL_REGIONS, so that the few blocks of a small launch (class 2) are not confined to
	 * a fraction of the pool; a state word names 2^26 chunks of its region */
	s.sc.n_regions = std::max<uint32_t>(1, std::min<uint32_t>(POOL_REGIONS, grid / 8));
	s.sc.region_u4 = pool.bytes / s.sc.n_regions / 4096 * 256;
	s.sc.pool_cap = (uint32_t)std::min<size_t>(s.sc.region_u4 * 16 / (c->wide ? 2048 : 1024), (size_t)1 << 26);
	{ /* private runs take at most three quarters of a region */
		const uint32_t lanes = (grid + s.sc.n_regions - 1) / s.sc.n_regions * LANE_BLOCK;
		s.sc.keep = std::min<uint32_t>(c->keep, s.sc.pool_cap / 4 * 3 / lanes);
	}
	hipEvent_t e0 = get_event(c), e1 = get_event(c);
	if (!e0 || !e1) return fail(BWB_E_HIP, "hipEventCreate failed");
	c->pending.push_back(PendingTime{ e0, e1, 1 });
	HIPCHK(hipEventRecord(e0, c->stream));
	const SlotDesc *descs = c->d_descs.as<SlotDesc>();
	unsigned long long *st = c->d_stats.as<unsigned long long>();
#define LAUNCH_SEARCH(PT, W, M) hipLaunchKernelGGL((kl_search<PT, W, M>), dim3(grid), dim3(LANE_BLOCK), lds, c->stream, c->ix, descs, wk, c->kp, s.sc, st)
	const bool multi = c->kp.multiref != 0;
	if (c->pos32 && !c->wide) { if (multi) LAUNCH_SEARCH(uint32_t, false, true); else LAUNCH_SEARCH(uint32_t, false, false); }
	else if (c->pos32) { if (multi) LAUNCH_SEARCH(uint32_t, true, true); else LAUNCH_SEARCH(uint32_t, true, false); }
	else if (!c->wide) { if (multi) LAUNCH_SEARCH(uint64_t, false, true); else LAUNCH_SEARCH(uint64_t, false, false); }
	else { if (multi) LAUNCH_SEARCH(uint64_t, true, true); else LAUNCH_SEARCH(uint64_t, true, false); }
#undef LAUNCH_SEARCH
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(e1, c->stream));
	if (k == 0) {
		c->parked = suspend || slice_iters != 0;
		hipEvent_t le = get_event(c);
		if (!le) return fail(BWB_E_HIP, "hipEventCreate failed");
		HIPCHK(hipEventRecord(le, c->stream));
		c->launch_ev.push_back(le);
		c->n_launches++;
		/* slot_wait starts at the slot's own launch: events of launches before the oldest slot in flight are never needed again */
		uint64_t oldest = c->n_launches;
		for (const Slot &t : c->slots) if (t.submitted && !t.complete && t.launch) oldest = std::min(oldest, t.launch);
		size_t drop = 0;
		while (c->ev_base + drop + 1 < oldest && drop < c->launch_ev.size()) c->free_events.push_back(c->launch_ev[drop++]);
		if (drop) { c->launch_ev.erase(c->launch_ev.begin(), c->launch_ev.begin() + (long)drop); c->ev_base += drop; }
	}
	if (c->dbg) {
		HIPCHK(hipStreamSynchronize(c->stream));
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, e0, e1));
		unsigned int hb[POOL_REGIONS * 16], used_max = 0;
		unsigned long long used = 0;
		int rc = fetch(c, hb, bump, sizeof(hb));
		if (rc) return rc;
		const uint32_t priv = (grid + s.sc.n_regions - 1) / s.sc.n_regions * LANE_BLOCK * s.sc.keep;
		for (uint32_t r = 0; r < s.sc.n_regions; r++) { used += std::min(priv + hb[r * 16], s.sc.pool_cap); used_max = std::max(used_max, priv + hb[r * 16]); }
		fprintf(stderr, "[bwb] kl_search class %d slot %d: %u new reads, grid %u, %s%s, %.3f ms, pool chunks used %llu of %d x %u (fullest region asked for %u; %u private per lane)\n",
		        k, si, n_work, grid, suspend ? "slice (parks)" : "drains", resume ? ", resumes parked reads" : "", ms, used, (int)s.sc.n_regions, s.sc.pool_cap, used_max, s.sc.keep);
	}
	if (c->launch_log) return log_launch(c, "kl_search", k, si, !(suspend || slice_iters != 0), c->stream, e0, e1);
	return BWB_OK;
}

/* kl_calc_d (+ the D_seed inheritance) of an uploaded slot on the second kernel stream, ahead of the slot's submit */
static int queue_calc_d(bwb_hip_ctx *c, int si) {
	Slot &s = c->slots[si];
	HIPCHK(s.ev_calcd.create());
	HIPCHK(hipStreamWaitEvent(c->dstream, s.ev_up.e, 0));
	HIPCHK(hipMemsetAsync(s.d_status.p, 0, (size_t)s.n_tot, c->dstream));
	int rc = launch_calc_d(c, 0, si, nullptr, s.n_tot, s.ctl_counter(), nullptr, nullptr, c->dstream);
	if (rc) return rc;
	rc = launch_inherit(c, si, c->dstream);
	if (rc) return rc;
	HIPCHK(hipEventRecord(s.ev_calcd.e, c->dstream));
	s.calcd_queued = true;
	return BWB_OK;
}

static int submit(bwb_hip_ctx *c, int si, bool suspend) {
	Slot &s = c->slots[si];
	if (!s.uploaded) return fail(BWB_E_STATE, "slot_submit: nothing uploaded into this slot");
	if (s.submitted && !s.complete) return fail(BWB_E_STATE, "slot_submit: the slot is still in flight");
	int rc = index_ready(c); /* (a context created with bwb_hip_ctx_create_async: its index upload may still be running) */
	if (rc) return rc;
	rc = ensure_class(c, 0);
	if (rc) return rc;
	rc = ensure_dtab(c, s.n_reads);
	if (rc) return rc;
	HIPCHK(hipStreamWaitEvent(c->stream, s.ev_up.e, 0));
	const bool ahead = c->calcd_ahead > 0 && s.n_reads != 0;
	if (ahead) { /* kl_calc_d of this batch runs (or has run) on the second kernel stream: the search waits for it (the slot's cursor is its cursor too) */
		if (!s.calcd_queued) { rc = queue_calc_d(c, si); if (rc) return rc; }
		HIPCHK(hipStreamWaitEvent(c->stream, s.ev_calcd.e, 0));
	}
	HIPCHK(hipMemsetAsync(s.d_ctl.p, 0, 256, c->stream));
	s.submitted = true; s.complete = false; s.fetched = false; s.launch = 0;
	if (s.n_reads == 0) { s.complete = true; s.launch = c->n_launches; return BWB_OK; }
	HIPCHK(hipMemsetAsync(s.d_n.p, 0, (size_t)s.n_reads * 4, c->stream));
	if (!ahead) {
		HIPCHK(hipMemsetAsync(s.d_status.p, 0, (size_t)s.n_tot, c->stream));
		rc = launch_calc_d(c, 0, si, nullptr, s.n_tot, s.ctl_counter(), nullptr, nullptr);
		if (rc) return rc;
		rc = launch_inherit(c, si);
		if (rc) return rc;
	}
	rc = launch_search(c, 0, si, nullptr, s.n_reads, s.ctl_counter(), suspend);
	if (rc) return rc;
	s.launch = c->n_launches;
	/* the batches the caller has uploaded beyond this one (slots are used round-robin): their kl_calc_d is queued NOW, on the second
	 * stream, where it runs beside the search slices instead of in front of its own (the two calls are independent per read,
	 * inexact_match.c:140-144, and kl_calc_d has its own lists) */
	for (int k = 1; ahead && k <= c->calcd_ahead; k++) {
		const int sj = (si + k) % BWB_MAX_SLOTS;
		Slot &t = c->slots[sj];
		if (t.uploaded && !t.submitted && !t.calcd_queued && t.n_reads) { rc = queue_calc_d(c, sj); if (rc) return rc; }
	}
	return BWB_OK;
}

extern "C" int bwb_hip_slot_submit(bwb_hip_ctx *c, int si) {
	if (!c || si < 0 || si >= BWB_MAX_SLOTS) return fail(BWB_E_ARG, "slot_submit: bad argument");
	HIPCHK(hipSetDevice(c->device));
	return submit(c, si, true);
}

/* reads of the slot whose status == want -> the slot's device worklist */
static int collect(bwb_hip_ctx *c, Slot &s, uint8_t want, std::vector<uint32_t> &ids) {
	s.h_status.resize(s.n_reads);
	int rc = fetch(c, s.h_status.data(), s.d_status.p, s.n_reads);
	if (rc) return rc;
	ids.clear();
	for (uint32_t i = 0; i < s.n_reads; i++) if (s.h_status[i] == want) ids.push_back(i);
	if (!ids.empty()) {
		HIPCHK(hipMemcpyAsync(s.d_worklist.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	return BWB_OK;
}

/* grows the slot's hit log (keeping what is in it) */
static int grow_log(bwb_hip_ctx *c, int si) {
	Slot &s = c->slots[si];
	unsigned long long cnt = 0;
	int rc = fetch(c, &cnt, s.ctl_count(), 8);
	if (rc) return rc;
	const uint64_t valid = std::min<uint64_t>(cnt, s.log_cap);
	const uint64_t ncap = s.log_cap * 4;
	DevMem nl;
	HIPCHK(nl.alloc(ncap * sizeof(bwb_aln)));
	HIPCHK(hipMemcpyAsync(nl.p, s.d_log.p, valid * sizeof(bwb_aln), hipMemcpyDeviceToDevice, c->stream));
	cnt = valid;
	HIPCHK(hipMemcpyAsync(s.ctl_count(), &cnt, 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	std::swap(s.d_log.p, nl.p); std::swap(s.d_log.bytes, nl.bytes);
	s.log_cap = ncap;
	c->h_descs[si].out.alns = s.d_log.as<uint4>(); c->h_descs[si].out.cap = ncap;
	HIPCHK(hipMemcpyAsync(c->d_descs.as<SlotDesc>() + si, &c->h_descs[si], sizeof(SlotDesc), hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return BWB_OK;
}

/* Re-runs, in the larger scratch classes, the reads of a finished slot that did not fit class 0: an SA-interval list or the
 * hit list overflowed, the chunk pool was empty when the read needed it, or the slot's hit log was full.  Draining launches
 * on the kernel stream (behind whatever slices are queued there). */
static int rerun_overflows(bwb_hip_ctx *c, int si) {
	Slot &s = c->slots[si];
	std::vector<uint32_t> todo, dids, list;
	/* The re-run classes take the whole chunk pool (a read that found it empty among 131 072 others gets, in class 1, four times
	 * its share and in class 2 a five-hundredth of the pool to itself).  So whatever is parked must finish first: one draining
	 * launch (fed from this slot's exhausted cursor) completes the parked reads of every slot. */
	if (c->parked) {
		int rc = launch_search(c, 0, si, nullptr, s.n_reads, s.ctl_counter(), false, false, true);
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(c->stream));
	}
	/* (n_tot: a carried read that rides along for its D_seed can overflow calculate_d too; it is never searched) */
	auto load_status = [&]() { s.h_status.resize(s.n_tot); return fetch(c, s.h_status.data(), s.d_status.p, s.n_tot); };
	auto put_worklist = [&](const std::vector<uint32_t> &v) -> int {
		HIPCHK(hipMemcpyAsync(s.d_worklist.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIPCHK(hipStreamSynchronize(c->stream));
		return BWB_OK;
	};
	for (int k = 1; k <= 2; k++) {
		int rc = load_status();
		if (rc) return rc;
		todo.clear(); dids.clear();
		for (uint32_t i = 0; i < s.n_tot; i++) {
			if (s.h_status[i] != ST_OK && i < s.n_reads) todo.push_back(i);
			if (s.h_status[i] == ST_D_OVF) dids.push_back(i);
		}
		if (todo.empty()) return BWB_OK;
		if (k == 1) c->stats.n_overflow_reads += todo.size(); /* (class 2 re-runs a subset of these) */
		rc = ensure_class(c, k);
		if (rc) return rc;
		if (!dids.empty()) { /* calculate_d first: it leaves ST_OK, or ST_D_OVF again (then the next class tries) */
			rc = put_worklist(dids);
			if (rc) return rc;
			rc = launch_calc_d(c, k, si, s.d_worklist.as<uint32_t>(), (uint32_t)dids.size(), s.ctl_counter2(), nullptr, nullptr);
			if (rc) return rc;
			rc = launch_inherit(c, si); /* sources that were late are there now (a read whose source still is not stays ST_D_OVF) */
			if (rc) return rc;
			HIPCHK(hipStreamSynchronize(c->stream));
			rc = load_status();
			if (rc) return rc;
		}
		list.clear();
		for (uint32_t i : todo) if (s.h_status[i] != ST_D_OVF) list.push_back(i);
		for (int guard = 0; !list.empty(); guard++) {
			if (guard == 40) return fail(BWB_E_OVERFLOW, "hit log kept overflowing");
			bool out_ovf = false;
			for (uint32_t i : list) out_ovf |= s.h_status[i] == ST_OUT_OVF;
			if (out_ovf) { rc = grow_log(c, si); if (rc) return rc; }
			rc = put_worklist(list);
			if (rc) return rc;
			rc = launch_search(c, k, si, s.d_worklist.as<uint32_t>(), (uint32_t)list.size(), s.ctl_counter2(), false);
			if (rc) return rc;
			HIPCHK(hipStreamSynchronize(c->stream));
			rc = load_status();
			if (rc) return rc;
			/* only reads the hit log had no room for are repeated within a class (the log grows x4 each time) */
			std::vector<uint32_t> again;
			for (uint32_t i : list) if (s.h_status[i] == ST_OUT_OVF) again.push_back(i);
			list.swap(again);
		}
	}
	int rc = load_status();
	if (rc) return rc;
	for (uint32_t i = 0; i < s.n_tot; i++)
		if (s.h_status[i] != ST_OK) return fail(BWB_E_OVERFLOW, "a read exceeded the largest per-read scratch class");
	return BWB_OK;
}

static int slot_wait(bwb_hip_ctx *c, int si) {
	Slot &s = c->slots[si];
	if (!s.submitted) return fail(BWB_E_STATE, "slot_wait: the slot has not been submitted");
	if (s.complete) return BWB_OK;
	/* A slice leaves the slot's unfinished reads parked; the slice after it resumes them and normally finishes them long
	 * before it ends.  So: wait for the slot's own slice; if reads are left, for the next queued one; and when nothing is
	 * queued any more, launch a draining slice. */
	uint64_t L = s.launch;
	for (;;) {
		HIPCHK(hipEventSynchronize(c->launch_ev[L - 1 - c->ev_base]));
		unsigned int done = 0;
		int rc = fetch(c, &done, s.ctl_done(), 4);
		if (rc) return rc;
		if (done >= s.n_reads) break;
		if (L < c->n_launches) { L++; continue; }
		/* (the slot's cursor is where the slices left it: normally exhausted, so only parked reads run) */
		rc = launch_search(c, 0, si, nullptr, s.n_reads, s.ctl_counter(), false, false);
		if (rc) return rc;
		L = c->n_launches;
	}
	int rc = resolve_times(c, false);
	if (rc) return rc;
	/* anything that needs a larger scratch class?  (one byte per read; almost always all zero) */
	s.h_status.resize(s.n_tot);
	rc = fetch(c, s.h_status.data(), s.d_status.p, s.n_tot);
	if (rc) return rc;
	bool clean = true;
	for (uint32_t i = 0; i < s.n_tot && clean; i++) clean = s.h_status[i] == ST_OK;
	if (!clean) { rc = rerun_overflows(c, si); if (rc) return rc; }
	s.complete = true;
	if (!any_in_flight(c)) c->parked = false; /* every read of every submitted slot is done: nothing can be parked any more */
	return BWB_OK;
}

extern "C" int bwb_hip_slot_wait(bwb_hip_ctx *c, int si) {
	if (!c || si < 0 || si >= BWB_MAX_SLOTS) return fail(BWB_E_ARG, "slot_wait: bad argument");
	HIPCHK(hipSetDevice(c->device));
	return slot_wait(c, si);
}

extern "C" int bwb_hip_slot_result(bwb_hip_ctx *c, int si, bwb_result *out) {
	if (!c || !out || si < 0 || si >= BWB_MAX_SLOTS) return fail(BWB_E_ARG, "slot_result: bad argument");
	HIPCHK(hipSetDevice(c->device));
	Slot &s = c->slots[si];
	if (!s.submitted) return fail(BWB_E_STATE, "slot_result: the slot has not been submitted");
	int rc = slot_wait(c, si);
	if (rc) return rc;
	const uint32_t n = s.n_reads;
	if (!s.fetched) {
		/* per-read counts and offsets into the hit log, and the log itself, on the result stream (the kernel stream keeps running) */
		unsigned long long cnt = 0;
		if (n) {
			HIPCHK(s.h_cnt.reserve((size_t)n * 4));
			HIPCHK(s.h_off.reserve((size_t)n * 8));
			HIPCHK(hipMemcpyAsync(s.h_cnt.p, s.d_n.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->rstream));
			HIPCHK(hipMemcpyAsync(s.h_off.p, s.d_off.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->rstream));
			HIPCHK(hipMemcpyAsync(s.h_ctl.p, s.ctl_count(), 8, hipMemcpyDeviceToHost, c->rstream));
			HIPCHK(hipStreamSynchronize(c->rstream));
			cnt = std::min<unsigned long long>(*s.h_ctl.as<unsigned long long>(), s.log_cap);
			if (cnt) {
				HIPCHK(s.h_log.reserve((size_t)cnt * sizeof(bwb_aln)));
				HIPCHK(hipMemcpyAsync(s.h_log.p, s.d_log.p, (size_t)cnt * sizeof(bwb_aln), hipMemcpyDeviceToHost, c->rstream));
				HIPCHK(hipStreamSynchronize(c->rstream));
			}
		}
		/* into read order (the log is in completion order; a read's hits are contiguous and in discovery order) */
		const uint32_t *hc = s.h_cnt.as<uint32_t>();
		const uint64_t *ho = s.h_off.as<uint64_t>();
		const bwb_aln *hl = s.h_log.as<bwb_aln>();
		s.h_aln_off.assign((size_t)n + 1, 0);
		for (uint32_t i = 0; i < n; i++) s.h_aln_off[i + 1] = s.h_aln_off[i] + hc[i];
		const uint64_t total = s.h_aln_off[n];
		s.h_alns.resize(total ? total : 1);
		for (uint32_t i = 0; i < n; i++) {
			if (!hc[i]) continue;
			if (ho[i] + hc[i] > cnt) return fail(BWB_E_STATE, "slot_result: a read's hits lie outside the hit log");
			memcpy(&s.h_alns[s.h_aln_off[i]], hl + ho[i], (size_t)hc[i] * sizeof(bwb_aln));
		}
		s.fetched = true;
	}
	out->n_reads = n;
	out->aln_off = s.h_aln_off.data();
	out->alns = s.h_alns.data();
	return BWB_OK;
}

/* ---- the one-batch interface: slot 0, one draining launch ------------------------------------------------------ */
extern "C" int bwb_hip_batch_upload(bwb_hip_ctx *c, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                                    uint32_t n_reads, uint32_t stride) {
	if (!c) return fail(BWB_E_ARG, "batch_upload: null context");
	int rc = bwb_hip_flush(c); /* this interface owns the context: nothing else may be in flight */
	if (rc) return rc;
	rc = bwb_hip_slot_upload(c, 0, p, reads_fwd, lens, n_reads, stride, nullptr, 0);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(c->cstream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return BWB_OK;
}

static int read_device_stats(bwb_hip_ctx *c) {
	unsigned long long st[STAT_WORDS];
	int rc = fetch(c, st, c->d_stats.p, sizeof(st));
	if (rc) return rc;
	c->stats.visits_single = st[STAT_VIS_SINGLE]; c->stats.visits_alphabet = st[STAT_VIS_ALPHA]; c->stats.visits_calc_d = st[STAT_VIS_CALCD];
	c->stats.heap_pops = st[STAT_POPS]; c->stats.heap_pushes = st[STAT_PUSHES]; c->stats.n_alignments = st[STAT_ALNS];
	c->stats.bucket_loads_search = st[STAT_BKT_SEARCH]; c->stats.bucket_loads_calc_d = st[STAT_BKT_CALCD];
	c->stats.n_parked_reads = st[STAT_PARKED];
	c->stats.lane_iterations = st[STAT_N]; c->stats.wave_iterations = st[STAT_WAVE_ITERS];
	c->stats.heap_entries_stored = st[STAT_ENT_ST]; c->stats.heap_entries_loaded = st[STAT_ENT_LD]; c->stats.record_loads = st[STAT_REC_LD];
#ifdef BWB_HIST
	{
		static const char *hn[H_N] = { "iter", "pop", "pop_from_mirror", "pop_gapped", "pruned", "hit", "exact_start", "expand", "exact_step", "need_rank", "same_bkt", "two_bkt",
			"w1", "w2", "w3_4", "w5_8", "w9_32", "w33_128", "w_big", "ne0", "ne1", "ne2", "ne3_4", "ne5_8", "ne9+", "push_gap", "push_mis", "push_match",
			"del_ok", "mm_ok", "ins_ok", "top_reload", "wave_iters", "wave_gaploop_trips", "wave_misloop_trips", "wave_matchloop_trips", "wave_any_two_bkt", "wave_any_wide8",
			"wave_nreq_le16", "alpha", "exact_multi", "finish", "group_pops", "wave_any_exact", "wave_any_expand", "wave_all_exact", "exp_same_w1", "exp_same_w2_4", "count_only_gap", "count_only_mis", "count_only_pop", "expand_after_hit" };
		fprintf(stderr, "[bwb hist]");
		for (int k = 0; k < H_N; k++) fprintf(stderr, " %s=%llu", hn[k], st[STAT_HIST + k]);
		fprintf(stderr, "\n");
	}
#endif
#ifdef BWB_BBPROF
	if (const char *bo = getenv("BWB_BBPROF_OUT")) { /* tools/bbprof.py: the basic-block counters of the instrumented kernels, one JSON line per call */
		if (FILE *f = fopen(bo, "a")) {
			fprintf(f, "{\"wave_iterations_search\": %llu, \"wave_iterations_calc_d\": %llu, \"lane_iterations_search\": %llu, \"counters\": [", st[STAT_WAVE_ITERS], st[STAT_WAVE_ITERS_CALCD], st[STAT_N]);
			for (int k = 0; k < 2048; k++) fprintf(f, "%s%llu", k ? "," : "", st[STAT_BBPROF + k]);
			fprintf(f, "]}\n");
			fclose(f);
		}
	}
#endif
#ifdef BWB_STAMPS
	if (st[STAT_STAMPS + 3]) {
		double tot = 0; for (int k = 0; k < 16; k++) tot += (double)st[STAT_STAMPS + k];
		const char *nm[16] = { "top", "A(pop)", "B(issue)", "C(rank)", "D.tail", "E.top-reload", "E.exact-step", "F(finish,grab)", "D.prune/hit", "D.masks", "D.reserve", "D.templates", "D.gap", "D.mm/match", "C.gather(exchange,loads,wait)", "E.exact-done/hits" };
		fprintf(stderr, "[bwb] stamps (%% of lane cycles):");
		for (int k = 0; k < 16; k++) if (st[STAT_STAMPS + k]) fprintf(stderr, " %s %.1f |", nm[k], 100.0 * (double)st[STAT_STAMPS + k] / tot);
		fprintf(stderr, "\n");
	}
#endif
	if (c->dbg) {
		fprintf(stderr, "[bwb] search loop iterations: total %llu, wave iterations %llu (%.1f of 64 lanes busy), reads parked at slice ends %llu\n",
		        st[STAT_N], st[STAT_WAVE_ITERS], st[STAT_WAVE_ITERS] ? (double)st[STAT_N] / (double)st[STAT_WAVE_ITERS] : 0.0, st[STAT_PARKED]);
	}
	return BWB_OK;
}

extern "C" int bwb_hip_reset_stats(bwb_hip_ctx *c) {
	if (!c) return fail(BWB_E_ARG, "reset_stats: null context");
	HIPCHK(hipSetDevice(c->device));
	int rc = resolve_times(c, false);
	if (rc) return rc;
	memset(&c->stats, 0, sizeof(c->stats));
	memset(c->log_prev, 0, sizeof(c->log_prev));
	HIPCHK(hipMemsetAsync(c->d_stats.p, 0, sizeof(unsigned long long) * STAT_WORDS, c->stream));
	c->t_run0 = wall_s();
	return BWB_OK;
}

extern "C" int bwb_hip_batch_run(bwb_hip_ctx *c) {
	if (!c || !c->slots[0].uploaded) return fail(BWB_E_STATE, "batch_run: no batch uploaded");
	HIPCHK(hipSetDevice(c->device));
	int rc = bwb_hip_flush(c);
	if (rc) return rc;
	rc = bwb_hip_reset_stats(c);
	if (rc) return rc;
	const double t0 = wall_s();
	rc = submit(c, 0, c->force_slices);
	if (rc) return rc;
	rc = slot_wait(c, 0);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(c->stream));
	c->stats.ms_total = (wall_s() - t0) * 1e3;
	rc = resolve_times(c, true);
	if (rc) return rc;
	return read_device_stats(c);
}

/* totals since the last reset_stats / batch_run; kernel times cover the launches that have finished */
extern "C" int bwb_hip_get_stats(bwb_hip_ctx *c, bwb_stats *out) {
	if (!c || !out) return fail(BWB_E_ARG, "get_stats: null argument");
	HIPCHK(hipSetDevice(c->device));
	int rc = resolve_times(c, false);
	if (rc) return rc;
	rc = read_device_stats(c);
	if (rc) return rc;
	*out = c->stats;
	return BWB_OK;
}

extern "C" int bwb_hip_batch_result(bwb_hip_ctx *c, bwb_result *out) {
	if (!c || !out) return fail(BWB_E_ARG, "batch_result: null argument");
	if (!c->slots[0].submitted || !c->slots[0].complete) return fail(BWB_E_STATE, "batch_result: batch_run has not completed");
	return bwb_hip_slot_result(c, 0, out);
}

extern "C" int bwb_hip_align_batch(bwb_hip_ctx *c, const bwb_params *p, const uint8_t *reads_fwd, const uint16_t *lens,
                                   uint32_t n_reads, uint32_t stride, bwb_result *out) {
	if (!c) return fail(BWB_E_ARG, "align_batch: null context");
	double t0 = wall_s();
	int rc = bwb_hip_batch_upload(c, p, reads_fwd, lens, n_reads, stride);
	if (rc) return rc;
	const double t1 = wall_s();
	rc = bwb_hip_batch_run(c);
	if (rc) return rc;
	const double t2 = wall_s();
	rc = bwb_hip_batch_result(c, out);
	if (c->dbg) fprintf(stderr, "[bwb] align_batch: upload %.3f s, run %.3f s, result %.3f s\n", t1 - t0, t2 - t1, wall_s() - t2);
	return rc;
}

extern "C" int bwb_hip_calc_d(bwb_hip_ctx *c, int32_t *out_D, int32_t *out_Dseed) {
	if (!c || !out_D || !out_Dseed) return fail(BWB_E_ARG, "calc_d: null argument");
	Slot &s = c->slots[0];
	if (!s.uploaded) return fail(BWB_E_STATE, "calc_d: no batch uploaded");
	HIPCHK(hipSetDevice(c->device));
	int rc = bwb_hip_flush(c);
	if (rc) return rc;
	rc = index_ready(c);
	if (rc) return rc;
	const size_t nD = (size_t)s.n_reads * (s.maxlen + 1) * 2, nS = (size_t)s.n_reads * (c->kp.seed_length + 1) * 2;
	DevMem dD, dS;
	HIPCHK(dD.alloc((nD ? nD : 1) * 4));
	HIPCHK(dS.alloc((nS ? nS : 1) * 4));
	HIPCHK(hipMemsetAsync(dD.p, 0, (nD ? nD : 1) * 4, c->stream));
	HIPCHK(hipMemsetAsync(dS.p, 0, (nS ? nS : 1) * 4, c->stream));
	rc = bwb_hip_reset_stats(c);
	if (rc) return rc;
	if (s.n_reads) {
		HIPCHK(hipStreamWaitEvent(c->stream, s.ev_up.e, 0));
		HIPCHK(hipMemsetAsync(s.d_status.p, 0, s.n_reads, c->stream));
		rc = launch_calc_d(c, 0, 0, nullptr, s.n_reads, s.ctl_counter(), dD.as<int32_t>(), dS.as<int32_t>());
		if (rc) return rc;
		HIPCHK(hipStreamSynchronize(c->stream));
		std::vector<uint32_t> ids;
		for (int k = 1; k <= 2; k++) {
			rc = collect(c, s, ST_D_OVF, ids);
			if (rc) return rc;
			if (ids.empty()) break;
			rc = ensure_class(c, k);
			if (rc) return rc;
			rc = launch_calc_d(c, k, 0, s.d_worklist.as<uint32_t>(), (uint32_t)ids.size(), s.ctl_counter2(), dD.as<int32_t>(), dS.as<int32_t>());
			if (rc) return rc;
			HIPCHK(hipStreamSynchronize(c->stream));
		}
		rc = collect(c, s, ST_D_OVF, ids);
		if (rc) return rc;
		if (!ids.empty()) return fail(BWB_E_OVERFLOW, "calculate_d: SA-interval list exceeded the largest scratch class");
	}
	HIPCHK(hipMemcpyAsync(out_D, dD.p, nD * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipMemcpyAsync(out_Dseed, dS.p, nS * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	rc = resolve_times(c, true);
	if (rc) return rc;
	return read_device_stats(c);
}

extern "C" int bwb_hip_rank16(bwb_hip_ctx *c, const uint64_t *pos, size_t n, int inc, int exact, uint64_t *out) {
	if (!c || (n && (!pos || !out))) return fail(BWB_E_ARG, "rank16: null argument");
	if (n == 0) return BWB_OK;
	for (size_t i = 0; i < n; i++)
		if (pos[i] != ~0ull && pos[i] >= c->ix.length) return fail(BWB_E_ARG, "rank16: position out of range");
	HIPCHK(hipSetDevice(c->device));
	{ int rc = index_ready(c); if (rc) return rc; }
	DevMem dp, dout;
	HIPCHK(dp.alloc(n * 8));
	HIPCHK(dout.alloc(n * 128));
	HIPCHK(hipMemcpyAsync(dp.p, pos, n * 8, hipMemcpyHostToDevice, c->stream));
	const unsigned grid = (unsigned)std::min<size_t>((n + BWB_OCTS_PER_BLOCK - 1) / BWB_OCTS_PER_BLOCK, (size_t)c->num_cu * 8);
	hipLaunchKernelGGL(k_rank16, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, dp.as<uint64_t>(), (uint64_t)n, inc, exact, dout.as<uint64_t>());
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, dout.p, n * 128, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	return BWB_OK;
}

/* layout 0: octet-cooperative rank (k_rank_bench); 1: one query per lane, the layout of the alignment kernels */
static int rank_bench(bwb_hip_ctx *c, int layout, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum) {
	if (!c || n < 4 || iters < 1) return fail(BWB_E_ARG, "rank_bench: bad argument");
	n &= ~(size_t)3;
	HIPCHK(hipSetDevice(c->device));
	{ int rc = index_ready(c); if (rc) return rc; }
	unsigned long long *cs = c->d_misc.as<unsigned long long>();
	HIPCHK(hipMemsetAsync(cs, 0, 8, c->stream));
	const unsigned grid = (unsigned)(c->num_cu * 8);
	auto launch = [&](uint64_t sd) {
		if (layout == 0) {
			if (c->pos32) hipLaunchKernelGGL(k_rank_bench<uint32_t>, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, cs);
			else hipLaunchKernelGGL(k_rank_bench<uint64_t>, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, cs);
		} else {
			if (c->pos32) hipLaunchKernelGGL(k_rank_bench_lane<uint32_t>, dim3(grid), dim3(LANE_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, cs);
			else hipLaunchKernelGGL(k_rank_bench_lane<uint64_t>, dim3(grid), dim3(LANE_BLOCK), 0, c->stream, c->ix, (uint64_t)n, sd, cs);
		}
	};
	Event e0, e1;
	HIPCHK(e0.create()); HIPCHK(e1.create());
	launch(seed); /* warm-up */
	HIPCHK(hipMemsetAsync(cs, 0, 8, c->stream));
	HIPCHK(hipEventRecord(e0.e, c->stream));
	for (int i = 0; i < iters; i++)
		launch(seed + i);
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(e1.e, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0.e, e1.e));
	unsigned long long v = 0;
	int rc = fetch(c, &v, cs, 8);
	if (rc) return rc;
	if (ms_per_iter) *ms_per_iter = ms / iters;
	if (checksum) *checksum = v;
	return BWB_OK;
}
extern "C" int bwb_hip_rank_bench(bwb_hip_ctx *c, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum) {
	return rank_bench(c, 0, n, iters, seed, ms_per_iter, checksum);
}
extern "C" int bwb_hip_rank_bench_lane(bwb_hip_ctx *c, size_t n, int iters, uint64_t seed, double *ms_per_iter, uint64_t *checksum) {
	return rank_bench(c, 1, n, iters, seed, ms_per_iter, checksum);
}

extern "C" int bwb_hip_set_sa(bwb_hip_ctx *c, const uint64_t *SA, uint64_t num_sa) {
	if (!c || !SA || num_sa != (c->ix.length + 31) / 32) return fail(BWB_E_ARG, "set_sa: bad argument");
	HIPCHK(hipSetDevice(c->device));
	HIPCHK(c->d_SA.alloc(num_sa * 8));
	HIPCHK(hipMemcpyAsync(c->d_SA.p, SA, num_sa * 8, hipMemcpyHostToDevice, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	c->num_sa = num_sa;
	return BWB_OK;
}

extern "C" int bwb_hip_locate(bwb_hip_ctx *c, const uint64_t *rows, size_t n, uint64_t *out_pos) {
	if (!c || (n && (!rows || !out_pos))) return fail(BWB_E_ARG, "locate: null argument");
	if (!c->d_SA.p) return fail(BWB_E_STATE, "locate: sampled SA not uploaded (bwb_hip_set_sa)");
	if (n == 0) { c->locate_ms = 0; c->locate_steps = 0; c->locate_rows = 0; return BWB_OK; } /* (locate_stats reports THIS call) */
	for (size_t i = 0; i < n; i++) if (rows[i] >= c->ix.length) return fail(BWB_E_ARG, "locate: row out of range");
	HIPCHK(hipSetDevice(c->device));
	{ int rc = index_ready(c); if (rc) return rc; }
	DevMem dr, dout;
	HIPCHK(dr.alloc(n * 8));
	HIPCHK(dout.alloc(n * 8));
	HIPCHK(hipMemcpyAsync(dr.p, rows, n * 8, hipMemcpyHostToDevice, c->stream));
	const unsigned grid = (unsigned)std::min<size_t>((n + BWB_OCTS_PER_BLOCK - 1) / BWB_OCTS_PER_BLOCK, (size_t)c->num_cu * 8);
	hipEvent_t e0 = get_event(c), e1 = get_event(c);
	struct EvGuard { /* the two events go back to the context's free list on every way out (an error path used to leak them) */
		bwb_hip_ctx *c; hipEvent_t a, b;
		~EvGuard() { if (a) c->free_events.push_back(a); if (b) c->free_events.push_back(b); }
	} evg{ c, e0, e1 };
	if (!e0 || !e1) return fail(BWB_E_HIP, "hipEventCreate failed");
	unsigned long long *steps = c->d_stats.as<unsigned long long>() + STAT_LOCATE_STEPS;
	HIPCHK(hipMemsetAsync(steps, 0, 8, c->stream));
	HIPCHK(hipEventRecord(e0, c->stream));
	hipLaunchKernelGGL(k_locate, dim3(grid), dim3(BWB_BLOCK), 0, c->stream, c->ix, c->d_SA.as<uint64_t>(), c->sa0_index, dr.as<uint64_t>(), (uint64_t)n, dout.as<uint64_t>(), c->d_stats.as<unsigned long long>());
	HIPCHK(hipGetLastError());
	HIPCHK(hipEventRecord(e1, c->stream));
	HIPCHK(hipMemcpyAsync(out_pos, dout.p, n * 8, hipMemcpyDeviceToHost, c->stream));
	unsigned long long hsteps = 0;
	HIPCHK(hipMemcpyAsync(&hsteps, steps, 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(hipStreamSynchronize(c->stream));
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, e0, e1));
	c->locate_ms = ms; c->locate_steps = hsteps; c->locate_rows = n;
	return BWB_OK;
}

/* the last bwb_hip_locate call: rows, invPsi steps (= rank-block visits: one 128-byte bucket each) and the kernel's HIP-event time */
extern "C" int bwb_hip_locate_stats(bwb_hip_ctx *c, uint64_t *rows, uint64_t *steps, double *kernel_ms) {
	if (!c) return fail(BWB_E_ARG, "locate_stats: null context");
	if (rows) *rows = c->locate_rows;
	if (steps) *steps = c->locate_steps;
	if (kernel_ms) *kernel_ms = c->locate_ms;
	return BWB_OK;
}

/* developer aid (not in include/bwbble_hip.h): per-read loop iteration counts of slot 0 when BWB_DEBUG_ITERS is set */
extern "C" int bwb_hip_debug_iters(bwb_hip_ctx *c, uint32_t *out) {
	if (!c || !c->slots[0].d_dbg_iters.p) return fail(BWB_E_STATE, "debug iteration counts are off (set BWB_DEBUG_ITERS)");
	return fetch(c, out, c->slots[0].d_dbg_iters.p, (size_t)c->slots[0].n_reads * 4);
}

/* developer aid: kl_calc_d visits per read of slot 0 */
extern "C" int bwb_hip_debug_calcd_work(bwb_hip_ctx *c, uint32_t *out) {
	if (!c || !c->slots[0].uploaded) return fail(BWB_E_STATE, "no batch");
	Slot &s = c->slots[0];
	std::vector<uint8_t> h((size_t)s.n_reads * s.dstride);
	int rc = fetch(c, h.data(), s.d_dbuf.p, h.size());
	if (rc) return rc;
	for (uint32_t i = 0; i < s.n_reads; i++) memcpy(&out[i], &h[(size_t)i * s.dstride + s.dstride - 8], 4);
	return BWB_OK;
}
