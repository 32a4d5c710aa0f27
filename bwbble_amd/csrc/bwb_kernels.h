/*
 * bwb_kernels.h - shared kernel-side types and the utility kernels (gfx950).  Included by bwb_hip.hip.
 *
 *   k_relayout, k_relayout64   reference .bwt arrays -> 128-byte rank buckets of 64 characters (bwb_device.h)
 *   k_rank16     O_alphabet / O for a list of positions (parity tests), octet-cooperative rank
 *   k_rank_bench random-position rank micro-benchmark (octet-cooperative rank)
 *   k_locate     SA[row] by the invPsi walk (aln2sam)
 *   k_gather     per-read hit lists -> read order
 * The alignment kernels themselves (kl_calc_d, kl_search: one read per lane) are in bwb_lane.h.
 */
#pragma once
#include "bwb_device.h"

#define BWB_BLOCK 256
#define BWB_OCTS_PER_BLOCK 32
#define NONE32 0xFFFFFFFFu
#define BAD_LEN 0xFFFFu   /* lens[] marker of a read the kernels cannot represent (> 255 bases; < 12 with -P): it gets an empty record */
#define ST_OK 0
#define ST_SCRATCH_OVF 1  /* heap arena / interval list / hit list too small: re-run in a bigger class */
#define ST_OUT_OVF 2      /* the slot's hit log is full: host grows it and re-runs the read */
#define ST_D_OVF 3        /* kl_calc_d: SA-interval list too small; kl_search leaves the read to the re-run of both kernels in a bigger class */

struct KParams {
	int max_diff, max_gapo, max_gape, max_entries;
	int mm_score, gapo_score, gape_score;
	int seed_length, max_diff_seed, max_best, no_indel_length;
	int num_buckets;
	int use_precalc; /* -P: the heap starts from the exact matches of the last 12 bases of rc (inexact_match.c:269-279) */
	int multiref;   /* 0: single-genome mode (-S): 4-letter children A,G,C,T in rows 1..4, 1-to-1 exact matching */
};

/* one resident batch of reads (a "slot" of the context): inputs, what kl_calc_d hands to kl_search, per-read status */
struct Batch {
	const uint8_t *reads;     /* [n][stride] read->seq codes */
	const uint16_t *lens;
	uint32_t n_reads, stride;
	uint8_t *dbuf;            /* [n][dstride]: what kl_calc_d hands to kl_search, one 16-byte RECORD PER FOUR READ POSITIONS (rec_put / rec_get below),
	                             then the read's N count (dstride-4) and its calculate_d work (dstride-8) */
	uint32_t dstride;
	uint8_t *status;          /* per read */
	uint32_t *dbg_iters;      /* optional (BWB_DEBUG_ITERS): loop iterations spent on each read */
};

struct OutBuf {
	uint4 *alns;              /* the slot's hit log, 48-byte records (bwb_aln: three uint4) */
	unsigned long long *count;
	uint64_t cap;
	uint64_t *off;            /* per read: first record */
	uint32_t *n;              /* per read: #records */
};

/* what kl_search needs to know about a slot: several slots are alive at once, because a read that is parked at the end of
 * a slice (see Work) belongs to an earlier batch than the reads the next slice starts */
struct SlotDesc {
	Batch b;
	OutBuf out;
	unsigned int *done;       /* reads of the slot that have been finished by kl_search (any status) */
};

/* the work of one launch */
struct Work {
	const uint32_t *worklist; /* read ids to process (NULL = 0..n_work-1) */
	uint32_t n_work;
	uint32_t *counter;        /* work-stealing cursor */
	uint32_t slot;            /* the slot those reads belong to */
	uint32_t suspend;         /* kl_search: when the cursor runs out a wave parks the reads it is working on in the lanes' save
	                             area and leaves (the next launch resumes them) instead of draining with ever fewer busy lanes */
	uint32_t slice_iters;     /* kl_search, test knob: park after this many loop iterations of a wave as well (0 = off) */
	uint32_t resume;          /* kl_search: lanes look for a parked read in their save area first */
};

enum { STAT_VIS_SINGLE = 0, STAT_VIS_ALPHA, STAT_POPS, STAT_PUSHES, STAT_ALNS, STAT_N, STAT_N_MAX, STAT_VIS_CALCD,
       STAT_BKT_SEARCH /* 128-byte buckets fetched by kl_search */, STAT_BKT_CALCD, STAT_PARKED /* reads parked at the end of a slice */,
       STAT_ENT_ST /* heap entries kl_search stored */, STAT_ENT_LD /* ... loaded */, STAT_REC_LD /* per-position records it loaded */,
       STAT_WAVE_ITERS = 16, STAT_WAVE_ITERS_CALCD = 17, STAT_LOCATE_STEPS = 18 /* k_locate: invPsi steps */, STAT_STAMPS = 24, STAT_HIST = 40 /* BWB_HIST diagnostic build */, STAT_BBPROF = 104 /* tools/bbprof.py: one 64-bit counter per basic block */,
#ifdef BWB_BBPROF
       STAT_WORDS = 104 + 2048 };
#else
       STAT_WORDS = 104 };
#endif

template <typename P> struct Intv { P L, U; };

/* The per-position data of a read, as kl_calc_d leaves it for kl_search.  An entry at read position i (aln_entry_t.i) consults
 * D[i-1], D[i-2] (inexact_match.c:317,399-405), D_seed[si-1], D_seed[si-2] with si = i - (len - seed_length) (:321-328,408-415) and the
 * base seq[len - i] it extends with.  One byte per D value: min(num_diff, 127) | 0x80 when sa_intv_width equals the previous
 * position's (the only way the width is ever used, :402-403,411-412).  Record m (16 bytes) serves the FOUR positions i = 4m .. 4m+3:
 *   bytes 0..5    D[k]   for k = 4m-2 .. 4m+3        (k < 0: 0)
 *   bytes 6..7    the bases of i = 4m .. 4m+3, 4 bits each (i & 3 = nibble)
 *   bytes 8..13   DS[k]  for the same k, DS[k] = D_seed[k - (len - seed_length)] - the seed's bounds moved to the read's positions
 *   bytes 14..15  m (the record's tag: kl_search keeps the record in registers and checks it against i >> 2)
 * so consecutive records overlap by two D bytes, and a search that walks down a read - a popped entry's match child is at i - 1 and is
 * popped next in 55 % of the pops - loads a record once per four positions (round 3: an 8-byte record per position, 0.8 loads per loop
 * iteration, 70 % of the kernel's metadata read requests). */
#define REC_BYTES 16
__host__ __device__ __forceinline__ uint32_t rec_count(uint32_t len) { return (len >> 2) + 1u; }
/* the byte D[k] (arr = 0) or DS[k] (arr = 8), k >= -2: in record (k + 2) >> 2, and again in the record before it when it is one of a record's first two */
/* (nrec = rec_count(len): the byte of the read's LAST position k = len - 1 has no record of its own when len % 4 == 3 - no entry sits beyond
 * position len - and is only written into the record before, where the entries at position len look for it; round 4 wrote it one record past
 * the read's records, into the per-read tail that holds calculate_d's work counter: ADVICE r4) */
__device__ __forceinline__ void rec_put(uint8_t *recs, uint32_t nrec, int arr, int k, uint32_t v) {
	const int m = (k + 2) >> 2, t = (k + 2) & 3;
	if ((uint32_t)m < nrec) recs[REC_BYTES * m + arr + t] = (uint8_t)v;
	if (t < 2 && m > 0) recs[REC_BYTES * (m - 1) + arr + 4 + t] = (uint8_t)v;
}
__device__ __forceinline__ uint32_t rec_get(const uint8_t *recs, int arr, int k) { return recs[REC_BYTES * ((k + 2) >> 2) + arr + ((k + 2) & 3)]; }

/* io.h:29,109: read base c (A0 G1 C2 T3) is compatible with code j iff gray(c) & grayVal[j]; N(10) excluded
 * (nucl_bases_table io.h:102-106).  Bit j of the mask = code j is a member. */
/* -S: child row j = 1..4 is base j-1 (A G C T), i.e. code nt4_gray[j-1] = 15, 3, 7, 1 (io.h:108, bwt.c:440-463) */
__device__ __forceinline__ uint32_t single_mask_codes(int c) { return c == 0 ? 1u << 15 : (c == 1 ? 1u << 3 : (c == 2 ? 1u << 7 : 1u << 1)); }
__device__ __forceinline__ uint32_t member_mask(int c) {
	/* A {8,9,11,12,13,14,15}  G {2,3,4,5,11,12,13}  C {4,5,6,7,8,9,11}  T {1,2,5,6,9,13,14} */
	return c == 0 ? 0xFB00u : (c == 1 ? 0x383Cu : (c == 2 ? 0x0BF0u : 0x6266u));
}

template <typename P> __device__ __forceinline__ void load_base(P *s_base, const DevIndex &ix) {
	for (int t = threadIdx.x; t < BWB_BASE_ROWS * 16; t += BWB_BLOCK) s_base[t] = (P)ix.base[t >> 4][t & 15];
	__syncthreads();
}
/* the base table (exact counts: O(), bwt.c:348-372), then its superblock rows once more as O_alphabet sees them (bwt.c:423-437): the codes
 * 5, 9, 11, 13 - never counted there - have C[j] - 1, to which the rank adds 1 - [first char of the block == j] (bwb_lane.h: side_finish);
 * the special positions -1 and length-1 are exact in that view too: they use the first table's rows */
template <typename P> __device__ __forceinline__ void load_base2(P *s_base, const DevIndex &ix) {
	for (int t = threadIdx.x; t < BWB_BASE_ROWS * 16; t += BWB_BLOCK) {
		const int row = t >> 4, j = t & 15;
		const P v = (P)ix.base[row][j];
		s_base[t] = v;
		if (row < BWB_NSB_MAX) s_base[BWB_BASE_ROWS * 16 + t] = ((0x2A20u >> j) & 1u) ? (P)(ix.base[BWB_ROW_NEG][j] - 1) : v;
	}
	__syncthreads();
}

/* ============================================================================================
 * Index re-layout: reference arrays (bwt words io.c:590-609, O rows bwt.c:280-291) -> buckets
 * One thread per 16-byte slice. blk0 = first block of this chunk; bwt/O point at the chunk.
 * ========================================================================================== */
__global__ void k_relayout(const uint32_t *bwt, const uint64_t *O, uint64_t blk0, uint64_t nblk_chunk, uint64_t nwords_chunk, uint64_t sa0_index,
                           const uint64_t *sbcount /* [NSB][16] exclusive counts at superblock starts */, uint4 *buckets) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk_chunk * 8) return;
	const uint64_t lb = t >> 3, blk = blk0 + lb;
	const int sl = (int)(t & 7);
	const uint64_t w0 = lb * 16;
	uint4 o;
	if (sl < 4) {
		const uint32_t first = w0 < nwords_chunk ? (bwt[w0] >> 28) : 0u;
		const uint64_t *row = O + lb * 16;
		const uint64_t *sb = sbcount + (blk >> BWB_SB_SHIFT) * 16;
		const int cs[4] = { 2 * sl, 2 * sl + 1, 2 * sl + 8, 2 * sl + 9 };
		uint32_t v[4];
		for (int q = 0; q < 4; q++) {
			const int c = cs[q];
			/* O rows are inclusive of position 128k and skip the sentinel row (bwt.c:284-288) */
			const bool counted = first == (uint32_t)c && !(c == 0 && blk * 128 == sa0_index);
			v[q] = (uint32_t)(row[c] - (counted ? 1u : 0u) - sb[c]);
		}
		o = make_uint4(v[0], v[1], v[2], v[3]);
	} else {
		const int w = sl - 4;
		uint32_t p[4] = { 0, 0, 0, 0 };
		for (int q = 0; q < 4; q++) {
			const uint64_t wi = w0 + 4 * w + q;
			const uint32_t word = wi < nwords_chunk ? bwt[wi] : 0u;
			for (int n = 0; n < 8; n++) {
				const uint32_t code = (word >> (28 - 4 * n)) & 15u; /* first char in bits 31-28, io.c:597 */
				const int j = 8 * q + n;
				p[0] |= (code & 1u) << j; p[1] |= ((code >> 1) & 1u) << j; p[2] |= ((code >> 2) & 1u) << j; p[3] |= ((code >> 3) & 1u) << j;
			}
		}
		o = make_uint4(p[0], p[1], p[2], p[3]);
	}
	buckets[lb * 8 + sl] = o; /* (chunk-local: the 128-character form only lives in a staging buffer, k_relayout64 below makes the index) */
}

/* The index proper (bwb_device.h): 64-character buckets, derived from a chunk of 128-character ones.  One thread per new bucket. */
__device__ __forceinline__ void sub_counts16(const uint4 p, uint32_t out[16]) {
	const uint32_t a[4] = { ~p.x & ~p.y, p.x & ~p.y, ~p.x & p.y, p.x & p.y };
	const uint32_t b[4] = { ~p.z & ~p.w, p.z & ~p.w, ~p.z & p.w, p.z & p.w };
	for (int c = 0; c < 16; c++) out[c] = (uint32_t)__popc(a[c & 3] & b[c >> 2]);
}
__global__ void k_relayout64(const uint4 *b128, uint64_t nblk128, uint4 *b64) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk128 * 2) return;
	const uint4 *src = b128 + (t >> 1) * 8;
	const int half = (int)(t & 1);
	uint4 cnt[4] = { src[0], src[1], src[2], src[3] };
	const uint4 s4 = src[4];
	const uint4 pa = src[4 + 2 * half], pb = src[5 + 2 * half];
	const uint32_t first = (s4.x & 1u) | ((s4.y & 1u) << 1) | ((s4.z & 1u) << 2) | ((s4.w & 1u) << 3);
	uint32_t m0[16];
	sub_counts16(pa, m0);
	if (half) {
		uint32_t h0[16], h1[16];
		sub_counts16(src[4], h0); sub_counts16(src[5], h1);
		for (int s = 0; s < 4; s++) {
			cnt[s].x += h0[2 * s] + h1[2 * s]; cnt[s].y += h0[2 * s + 1] + h1[2 * s + 1];
			cnt[s].z += h0[2 * s + 8] + h1[2 * s + 8]; cnt[s].w += h0[2 * s + 9] + h1[2 * s + 9];
		}
	}
	uint32_t mid[4];
	for (int s = 0; s < 4; s++) mid[s] = m0[2 * s] | (m0[2 * s + 1] << 8) | (m0[2 * s + 8] << 16) | (m0[2 * s + 9] << 24);
	uint4 *dst = b64 + t * 8;
	dst[0] = cnt[0]; dst[1] = cnt[1]; dst[2] = cnt[2]; dst[3] = cnt[3];
	dst[4] = pa; dst[5] = pb;
	dst[6] = make_uint4(mid[0], mid[1], mid[2], mid[3]);
	dst[7] = make_uint4(first, 0u, 0u, 0u);
}

/* O_alphabet / exact Occ16 for a list of positions: one octet per query */
__global__ __launch_bounds__(BWB_BLOCK) void k_rank16(DevIndex ix, const uint64_t *pos, uint64_t n, int inc, int exact, uint64_t *out) {
	__shared__ uint64_t s_base[BWB_BASE_ROWS * 16];
	load_base<uint64_t>(s_base, ix);
	const int lane = threadIdx.x & 63, ol = lane & 7;
	const uint64_t noct = (uint64_t)gridDim.x * BWB_OCTS_PER_BLOCK;
	for (uint64_t q = (uint64_t)blockIdx.x * BWB_OCTS_PER_BLOCK + (threadIdx.x >> 3); q < n; q += noct) {
		RankReq<uint64_t> ra;
		rank_issue<uint64_t>(ix.buckets, ix.length - 1, pos[q], ol, ra);
		uint64_t v0, v1;
		if (exact) rank_finish<uint64_t, false>(ra, s_base, ol, lane, v0, v1);
		else rank_finish<uint64_t, true>(ra, s_base, ol, lane, v0, v1);
		out[q * 16 + 2 * ol] = ol == 0 ? 0 : v0 + inc;
		out[q * 16 + 2 * ol + 1] = v1 + inc;
	}
}

/* Rank micro-benchmark: pseudo-random positions, 4 independent visits in flight per octet */
template <typename P>
__global__ __launch_bounds__(BWB_BLOCK) void k_rank_bench(DevIndex ix, uint64_t n, uint64_t seed, unsigned long long *checksum) {
	__shared__ P s_base[BWB_BASE_ROWS * 16];
	load_base<P>(s_base, ix);
	const int lane = threadIdx.x & 63, ol = lane & 7;
	const uint64_t noct = (uint64_t)gridDim.x * BWB_OCTS_PER_BLOCK;
	unsigned long long acc = 0;
	for (uint64_t q = ((uint64_t)blockIdx.x * BWB_OCTS_PER_BLOCK + (threadIdx.x >> 3)) * 4; q < n; q += noct * 4) {
		RankReq<P> rq[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			uint64_t x = (q + u) * 0x9E3779B97F4A7C15ull + seed;
			x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
			rank_issue<P>(ix.buckets, (P)(ix.length - 1), (P)(x % (ix.length - 1)), ol, rq[u]);
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			P v0, v1;
			rank_finish<P, false>(rq[u], s_base, ol, lane, v0, v1);
			acc += (ol == 0 ? 0ull : (unsigned long long)v0) + 3ull * v1; /* codes 1..15 (code 0 = '$' is never a child) */
		}
	}
	acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
	if (ol == 0 && acc) atomicAdd(checksum, acc);
}

/* gathers the per-read hit lists into read order */
__global__ void k_gather(const uint4 *log, const uint64_t *off, const uint32_t *n, const uint64_t *dst_off, uint32_t n_reads, uint4 *dst) {
	const uint32_t rid = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);
	if (rid >= n_reads) return;
	const int ol = threadIdx.x & 7;
	const uint64_t so = off[rid] * 2, d = dst_off[rid] * 2;
	const uint32_t cnt = n[rid] * 2;
	for (uint32_t t = ol; t < cnt; t += 8) dst[d + t] = log[so + t];
}

/* D_seed of a read that is not longer than the seed.  The reference computes D_seed only when len > seed_length
 * (inexact_match.c:62-64) but inexact_match reads it regardless (:321-328,408-415): the serial path (-t 1, one buffer for the
 * whole file, :35) therefore sees the bounds of the LAST LONGER READ BEFORE IT in the file - zeros when there is none.
 * src[r] names that read (host: slot_upload); the records carry the seed's bounds at the read's own positions (rec_put), so read r
 * takes the byte of position k + len_q - len_r of read q.  Runs after kl_calc_d. */
__global__ void k_dseed_inherit(Batch b, const uint32_t *src, uint32_t n) {
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const uint32_t q = src[r];
	if (q == NONE32) return;
	if (b.status[q] == ST_D_OVF) { b.status[r] = ST_D_OVF; return; } /* its source waits for a larger scratch class: so does this read */
	const int lr = b.lens[r], lq = b.lens[q];
	uint8_t *rr = b.dbuf + (size_t)r * b.dstride;
	const uint8_t *rq = b.dbuf + (size_t)q * b.dstride;
	/* DS of read r at position k (of r) = DS of read q at position k + len_q - len_r (of q): the same seed index */
	for (int k = -2; k < lr; k++) rec_put(rr, rec_count((uint32_t)lr), 8, k, rec_get(rq, 8, k + lq - lr));
}

/* SA[row] by the invPsi walk (bwt.c:311-329): one octet per row */
__global__ __launch_bounds__(BWB_BLOCK) void k_locate(DevIndex ix, const uint64_t *SA, uint64_t sa0_index, const uint64_t *rows, uint64_t n, uint64_t *out, unsigned long long *stats) {
	__shared__ uint64_t s_base[BWB_BASE_ROWS * 16];
	load_base<uint64_t>(s_base, ix);
	const int lane = threadIdx.x & 63, ol = lane & 7;
	const uint64_t noct = (uint64_t)gridDim.x * BWB_OCTS_PER_BLOCK;
	for (uint64_t q = (uint64_t)blockIdx.x * BWB_OCTS_PER_BLOCK + (threadIdx.x >> 3); q < n; q += noct) {
		uint64_t i = rows[q], j = 0;
		while ((i & 31) != 0) { /* SA_INTERVAL = 32, bwt.h:16 */
			if (i == sa0_index) { i = 0; j++; continue; } /* invPsi bwt.c:312-314 */
			RankReq<uint64_t> ra;
			rank_issue<uint64_t>(ix.buckets, ix.length - 1, i, ol, ra);
			/* B(i), bwt.c:337-345: bit (i&31) of the planes held by lane 4 + ((i&63)>>5) */
			const uint4 cq = ra.regular ? ra.q : ix.buckets[(i >> BKT_SHIFT) * 8 + ol]; /* i == length-1 is not ranked via its bucket */
			const int off = (int)(i & BKT_MASK), srcl = (lane & ~7) + 4 + (off >> 5), bit = off & 31;
			const uint32_t code_here = ((cq.x >> bit) & 1u) | (((cq.y >> bit) & 1u) << 1) | (((cq.z >> bit) & 1u) << 2) | (((cq.w >> bit) & 1u) << 3);
			const uint32_t code = oct_bcast(code_here, srcl);
			uint64_t v0, v1;
			rank_finish<uint64_t, false>(ra, s_base, ol, lane, v0, v1);
			/* C[c] + O(c,i): held by lane c>>1.  The sentinel row is stored as code 0 but is not a '$' (bwt.c:364) */
			const uint64_t mine = (code & 1u) ? v1 : v0;
			uint64_t nxt = oct_bcast(mine, (lane & ~7) + (int)(code >> 1));
			if (code == 0 && ra.regular && sa0_index >= (i & ~127ull) && sa0_index <= i) nxt--;
			i = nxt;
			j++;
		}
		if (ol == 0) { out[q] = (SA[i >> 5] + j) % ix.length; if (j) atomicAdd(&stats[STAT_LOCATE_STEPS], (unsigned long long)j); } /* (j: invPsi steps = rank-block visits, the step through the sentinel row included) */
	}
}
