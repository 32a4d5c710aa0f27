/*
 * bwb_kernels.h - the alignment kernels (gfx950).  Included once by bwb_hip.hip.
 *
 * k_calc_d : calculate_d for the full read and for the seed (inexact_match.c:171-254, called at
 *            inexact_match.c:140-143) -> one byte per position (num_diff clamped to 127, bit 7 =
 *            "sa_intv_width equals the previous position's", the only way the width is ever used,
 *            inexact_match.c:402-403,411-412).
 * k_search : inexact_match (inexact_match.c:256-506) with exact_match_bounded (exact_match.c:66-119)
 *            as a mode of the same loop, so that every wave iteration of every octet is one rank
 *            visit pair.
 *
 * One read per OCTET (8 lanes); everything "uniform" below is replicated in the 8 lanes.
 * Kernels are templated on the position type P (uint32_t for BWT length < 2^32-1, else uint64_t).
 */
#pragma once
#include "bwb_device.h"

#define BWB_BLOCK 256
#define BWB_OCTS_PER_BLOCK 32
#define NONE32 0xFFFFFFFFu
#define ST_OK 0
#define ST_SCRATCH_OVF 1  /* heap arena / interval list / hit list too small: re-run in a bigger class */
#define ST_OUT_OVF 2      /* global hit buffer full: host grows it and re-runs the read */

struct KParams {
	int max_diff, max_gapo, max_gape, max_entries;
	int mm_score, gapo_score, gape_score;
	int seed_length, max_diff_seed, max_best, no_indel_length;
	int num_buckets;
};

struct Batch {
	const uint8_t *reads;     /* [n][stride] read->seq codes */
	const uint16_t *lens;
	uint32_t n_reads, stride;
	uint8_t *dbuf;            /* [n][dstride]: D bytes at 0, D_seed bytes at dseed_off */
	uint32_t dstride, dseed_off;
	const uint32_t *worklist; /* read ids to process (NULL = 0..n_work-1) */
	uint32_t n_work;
	uint32_t *counter;        /* work-stealing cursor */
	uint8_t *status;          /* per read */
	uint32_t *dbg_iters;      /* optional (BWB_DEBUG): loop iterations spent on each read */
};

struct Scratch {              /* per-octet private regions, slot = blockIdx.x*32 + octet */
	uint4 *ent;               /* [slots][nchunks*16*2]  heap entries, 32 B each */
	uint32_t *cprev;          /* [slots][nchunks]       chunk chain / free list */
	uint4 *alns;              /* [slots][acap*2]        hits of the current read */
	void *lists;              /* [slots][2*lcap]        SA-interval lists (cur/next) of Intv<P> */
	uint32_t nchunks, acap, lcap;
};

struct OutBuf {
	uint4 *alns;              /* global hit log, 32 B records */
	unsigned long long *count;
	uint64_t cap;
	uint64_t *off;            /* per read: first record */
	uint32_t *n;              /* per read: #records */
};

enum { STAT_VIS_SINGLE = 0, STAT_VIS_ALPHA, STAT_POPS, STAT_PUSHES, STAT_ALNS, STAT_N };

template <typename P> struct Intv { P L, U; };

/* io.h:29,109: read base c (A0 G1 C2 T3) is compatible with code j iff gray(c) & grayVal[j]; N(10) excluded
 * (nucl_bases_table io.h:102-106).  Bit j of the mask = code j is a member. */
__device__ __forceinline__ uint32_t member_mask(int c) {
	/* A {8,9,11,12,13,14,15}  G {2,3,4,5,11,12,13}  C {4,5,6,7,8,9,11}  T {1,2,5,6,9,13,14} */
	return c == 0 ? 0xFB00u : (c == 1 ? 0x383Cu : (c == 2 ? 0x0BF0u : 0x6266u));
}

template <typename P> __device__ __forceinline__ void load_base(P *s_base, const DevIndex &ix) {
	for (int t = threadIdx.x; t < BWB_BASE_ROWS * 16; t += BWB_BLOCK) s_base[t] = (P)ix.base[t >> 4][t & 15];
	__syncthreads();
}

/* ---------------------------------------------------------------------------------------------
 * add_sa_interval (align.c:93-110) for all children of one parent interval at once.
 * Lane ol holds the children for codes 2ol (L0,U0) and 2ol+1 (L1,U1); ne* = member && non-empty.
 * The list being built keeps its LAST interval ("open tail") in LDS (tail[0..1]) and the closed
 * ones in global memory (nlist[0..T-2]).  Returns the int-wrapped sum of child widths
 * (num_matches, inexact_match.c:227).
 * ------------------------------------------------------------------------------------------- */
template <typename P>
__device__ __forceinline__ int32_t append_children(P L0, P U0, P L1, P U1, bool ne0, bool ne1, int ol, int lane,
                                                   volatile P *tail, Intv<P> *nlist, int &T, int cap, bool &ovf) {
	const uint32_t m16 = oct_or(((uint32_t)ne0 | ((uint32_t)ne1 << 1)) << (2 * ol));
	uint32_t w = 0;
	if (ne0) w += (uint32_t)(U0 - L0 + 1);
	if (ne1) w += (uint32_t)(U1 - L1 + 1);
	w = oct_add(w);
	if (m16 == 0) return 0;
	const P tL = tail[0], tU = tail[1];
	const uint32_t below0 = m16 & ((1u << (2 * ol)) - 1u);
	const P lastU_mine = ne1 ? U1 : U0;
	const int p = below0 ? 31 - __clz((int)below0) : 0;
	const P carryU = oct_bcast(lastU_mine, (lane & ~7) + (p >> 1));
	const bool prev0 = below0 != 0 || T > 0;
	const P prevU0 = below0 ? carryU : tU;
	const bool new0 = ne0 && !(prev0 && L0 == (P)(prevU0 + 1));
	const bool prev1 = ne0 || prev0;
	const P prevU1 = ne0 ? U0 : prevU0;
	const bool new1 = ne1 && !(prev1 && L1 == (P)(prevU1 + 1));
	const uint32_t n16 = oct_or(((uint32_t)new0 | ((uint32_t)new1 << 1)) << (2 * ol));
	const int total_new = __popc(n16);
	const int newT = T + total_new;
	if (newT - 1 > cap) { ovf = true; return (int32_t)w; }
	const int firstk = __ffs((int)m16) - 1;
	const bool cont = T > 0 && !((n16 >> firstk) & 1u); /* the first child run extends the old tail */
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const bool ne = h ? ne1 : ne0, nw = h ? new1 : new0;
		const P L = h ? L1 : L0, U = h ? U1 : U0;
		const int k = 2 * ol + h;
		if (ne) {
			const uint32_t upto = (2u << k) - 1u;
			const int idx = T - 1 + __popc(n16 & upto);
			const uint32_t above = m16 & ~upto;
			const bool last = above == 0 || ((n16 >> (__ffs((int)above) - 1)) & 1u);
			if (idx == newT - 1) { /* the run that stays open */
				if (nw) tail[0] = L;
				if (last) tail[1] = U;
			} else {
				if (nw) nlist[idx].L = L;
				if (last) { nlist[idx].U = U; if (cont && idx == T - 1) nlist[idx].L = tL; }
			}
		}
	}
	/* old tail closed unchanged because the first child starts a new run */
	if (T > 0 && !cont && ol == (firstk >> 1)) { nlist[T - 1].L = tL; nlist[T - 1].U = tU; }
	T = newT;
	return (int32_t)w;
}

/* fetch next unit of work for an octet; returns read id or NONE32 */
__device__ __forceinline__ uint32_t next_read(const Batch &b, int ol, int lane) {
	uint32_t w = 0;
	if (ol == 0) w = atomicAdd(b.counter, 1u);
	w = oct_bcast(w, lane & ~7);
	if (w >= b.n_work) return NONE32;
	return b.worklist ? b.worklist[w] : w;
}

/* ============================================================================================
 * k_calc_d
 * ========================================================================================== */
template <typename P>
__global__ __launch_bounds__(BWB_BLOCK) void k_calc_d(DevIndex ix, Batch b, KParams kp, Scratch sc, int32_t *dbgD,
                                                      int32_t *dbgDs, uint32_t dbg_ld, uint32_t dbg_lds, unsigned long long *stats) {
	extern __shared__ __align__(16) unsigned char smem[];
	P *s_base = (P *)smem;                                                          /* BWB_BASE_ROWS*16 */
	volatile P *s_tail = (volatile P *)(smem + BWB_BASE_ROWS * 16 * 8);             /* [32][4] */
	volatile uint8_t *s_seq = (volatile uint8_t *)(smem + BWB_BASE_ROWS * 16 * 8 + BWB_OCTS_PER_BLOCK * 4 * 8); /* [32][spad] */
	const uint32_t spad = (b.stride + 15u) & ~15u;
	load_base<P>(s_base, ix);

	const int lane = threadIdx.x & 63, ol = lane & 7, ob = threadIdx.x >> 3;
	const uint32_t slot = blockIdx.x * BWB_OCTS_PER_BLOCK + ob;
	volatile P *tails = s_tail + ob * 4;
	volatile uint8_t *sseq = s_seq + ob * spad;
	Intv<P> *lbase = (Intv<P> *)sc.lists + (size_t)slot * 2 * sc.lcap;
	const int cap = (int)sc.lcap;
	const uint4 *__restrict__ buckets = ix.buckets;
	const P last_row = (P)(ix.length - 1);

	bool active = false, done = false;
	uint32_t rid = 0;
	int len = 0, phase = 0, plen = 0, r = 0, z = 0, s = 0, curT = 0, T = 0, cursel = 0;
	int32_t nm = 0, prev_nm = 0;
	unsigned long long vis = 0;

	for (;;) {
		if (!active && !done) {
			rid = next_read(b, ol, lane);
			if (rid == NONE32) done = true;
			else {
				len = b.lens[rid];
				for (int k = ol; k < len; k += 8) sseq[k] = b.reads[(size_t)rid * b.stride + k];
				phase = 0; plen = len; r = len - 1; z = 0; s = 0; T = 0; cursel = 0; nm = 0; prev_nm = 0;
				tails[0] = 0; tails[1] = last_row;
				curT = 1;
				active = len > 0;
				if (!active && ol == 0) b.status[rid] = ST_OK;
			}
		}
		if (__all(done)) break;
		if (active) {
			const int c = sseq[r];
			bool ovf = false;
			if (c <= 3) {
				/* interval s of the current list */
				P iL, iU;
				if (s == curT - 1) { iL = tails[cursel * 2]; iU = tails[cursel * 2 + 1]; }
				else { const Intv<P> v = (lbase + cursel * cap)[s]; iL = v.L; iU = v.U; }
				RankReq<P> ra, rb;
				rank_issue<P>(buckets, last_row, (P)(iL - 1), ol, ra);
				rank_issue<P>(buckets, last_row, iU, ol, rb);
				vis += (ra.regular ? 1 : 0) + (rb.regular ? 1 : 0);
				P a0, a1, u0, u1;
				rank_finish<P, false>(ra, s_base, ol, lane, a0, a1);
				rank_finish<P, false>(rb, s_base, ol, lane, u0, u1);
				const uint32_t mem = member_mask(c);
				const P L0 = a0 + 1, L1 = a1 + 1;
				const bool ne0 = ((mem >> (2 * ol)) & 1u) && (L0 <= u0);
				const bool ne1 = ((mem >> (2 * ol + 1)) & 1u) && (L1 <= u1);
				nm += append_children<P>(L0, u0, L1, u1, ne0, ne1, ol, lane, tails + (cursel ^ 1) * 2,
				                         lbase + (cursel ^ 1) * cap, T, cap, ovf);
				s++;
			}
			if (ovf) {
				if (ol == 0) b.status[rid] = ST_SCRATCH_OVF;
				active = false;
			} else if (c > 3 || s >= curT) {
				/* position finished: swap lists (inexact_match.c:234-237) */
				cursel ^= 1; curT = (c > 3) ? 0 : T; T = 0; s = 0;
				if (curT == 0) { /* no matches: restart with the full interval (inexact_match.c:240-244) */
					tails[cursel * 2] = 0; tails[cursel * 2 + 1] = last_row;
					curT = 1; z++;
					nm = (int32_t)(uint32_t)ix.length;
				}
				const int k = plen - 1 - r; /* D index */
				if (ol == 0) {
					const uint8_t byte = (uint8_t)((z > 127 ? 127 : z) | ((k > 0 && nm == prev_nm) ? 0x80 : 0));
					b.dbuf[(size_t)rid * b.dstride + (phase ? b.dseed_off : 0) + k] = byte;
					if (dbgD) {
						int32_t *dst = phase ? dbgDs + ((size_t)rid * dbg_lds + k) * 2 : dbgD + ((size_t)rid * dbg_ld + k) * 2;
						dst[0] = z; dst[1] = nm;
					}
				}
				prev_nm = nm; nm = 0; r--;
				if (r < 0) {
					if (ol == 0 && dbgD) { /* D[readLen] (inexact_match.c:249-250) */
						int32_t *dst = phase ? dbgDs + ((size_t)rid * dbg_lds + plen) * 2 : dbgD + ((size_t)rid * dbg_ld + plen) * 2;
						dst[0] = z + 1; dst[1] = 0;
					}
					if (phase == 0 && kp.seed_length && len > kp.seed_length) { /* inexact_match.c:141-143 */
						phase = 1; plen = kp.seed_length; r = plen - 1; z = 0; prev_nm = 0;
						tails[cursel * 2] = 0; tails[cursel * 2 + 1] = last_row;
						curT = 1;
					} else {
						if (ol == 0) b.status[rid] = ST_OK;
						active = false;
					}
				}
			}
		}
	}
	if (ol == 0 && vis) atomicAdd(&stats[STAT_VIS_SINGLE], vis);
}

/* ============================================================================================
 * k_search
 * ========================================================================================== */
#define MODE_POP 0
#define MODE_EXACT 1
#define STATE_M 0
#define STATE_I 1
#define STATE_D 2

/* Score-bucketed LIFO heap of the reference (inexact_match.h:17-34, inexact_match.c:510-610) as
 * per-bucket chains of 16-entry chunks inside the octet's private arena.  bstate[s] (LDS) =
 * (top chunk << 5) | fill, NONE32 when bucket s is empty. */
struct Heap {
	uint32_t bump, fhead;      /* never-used chunks start at bump; freed chunks chain from fhead via cprev */
	uint64_t neLo, neHi;       /* non-empty bucket bitmap (<=128 buckets) */
	int best, num_entries;
};
struct Resv { uint32_t c0, n1, n2; int f0; };

__device__ __forceinline__ uint32_t heap_alloc(Heap &h, uint32_t *cprev, uint32_t nchunks, bool &ovf) {
	uint32_t c = 0;
	if (h.bump < nchunks) c = h.bump++;
	else if (h.fhead != NONE32) { c = h.fhead; h.fhead = cprev[c]; }
	else ovf = true;
	return c;
}

/* make room for k (1..31) more entries on bucket s */
__device__ __forceinline__ Resv heap_reserve(Heap &h, int s, int k, volatile uint32_t *bstate, uint32_t *cprev,
                                             uint32_t nchunks, int ol, bool &ovf) {
	Resv r;
	const uint32_t bst = bstate[s];
	r.c0 = bst == NONE32 ? NONE32 : (bst >> 5);
	r.f0 = bst == NONE32 ? 16 : (int)(bst & 31u);
	r.n1 = r.n2 = NONE32;
	const int need = (r.f0 + k > 16) ? ((r.f0 + k - 16 + 15) >> 4) : 0;
	if (need >= 1) { r.n1 = heap_alloc(h, cprev, nchunks, ovf); if (!ovf && ol == 0) cprev[r.n1] = r.c0; }
	if (need >= 2) { r.n2 = heap_alloc(h, cprev, nchunks, ovf); if (!ovf && ol == 0) cprev[r.n2] = r.n1; }
	const uint32_t lastc = need == 0 ? r.c0 : (need == 1 ? r.n1 : r.n2);
	bstate[s] = (lastc << 5) | (uint32_t)(((r.f0 + k - 1) & 15) + 1);
	if (s < 64) h.neLo |= 1ull << s; else h.neHi |= 1ull << (s - 64);
	if (h.best > s) h.best = s;
	h.num_entries += k;
	return r;
}
__device__ __forceinline__ size_t resv_slot(const Resv &r, int t) {
	const int p = r.f0 + t;
	const uint32_t c = p < 16 ? r.c0 : (p < 32 ? r.n1 : r.n2);
	return (size_t)c * 16 + (p & 15);
}

template <typename P>
__global__ __launch_bounds__(BWB_BLOCK) void k_search(DevIndex ix, Batch b, KParams kp, Scratch sc, OutBuf out,
                                                      unsigned long long *stats, uint32_t lds_oct_bytes, uint32_t lpad,
                                                      uint32_t spadseed, uint32_t nbpad) {
	extern __shared__ __align__(16) unsigned char smem[];
	P *s_base = (P *)smem;
	load_base<P>(s_base, ix);

	const int lane = threadIdx.x & 63, ol = lane & 7, obk = threadIdx.x >> 3;
	const uint32_t slot = blockIdx.x * BWB_OCTS_PER_BLOCK + obk;
	/* per-octet LDS: tails[4] (32 B) | bstate[nbpad] u32 | D[lpad] | Dseed[spadseed] | rc[lpad] */
	unsigned char *my = smem + BWB_BASE_ROWS * 16 * 8 + (size_t)obk * lds_oct_bytes;
	volatile P *tails = (volatile P *)my;
	volatile uint32_t *bstate = (volatile uint32_t *)(my + 32);
	volatile uint8_t *sD = (volatile uint8_t *)(my + 32 + 4 * nbpad);
	volatile uint8_t *sDs = sD + lpad;
	volatile uint8_t *src = sDs + spadseed;

	uint4 *ent = sc.ent + (size_t)slot * sc.nchunks * 32;
	uint32_t *cprev = sc.cprev + (size_t)slot * sc.nchunks;
	uint4 *myalns = sc.alns + (size_t)slot * sc.acap * 2;
	Intv<P> *lbase = (Intv<P> *)sc.lists + (size_t)slot * 2 * sc.lcap;
	const int lcap = (int)sc.lcap;
	const int nb = kp.num_buckets;
	const uint4 *__restrict__ buckets = ix.buckets;
	const P last_row = (P)(ix.length - 1);

	bool active = false, done = false;
	uint32_t rid = 0;
	int len = 0, mode = MODE_POP;
	Heap h; h.bump = 0; h.fhead = NONE32; h.neLo = h.neHi = 0; h.best = nb; h.num_entries = 0;
	int best_score = 0, max_diff = 0, num_best = 0, n_alns = 0;
	int r = 0, s = 0, curT = 0, T = 0, cursel = 0;               /* exact-tail state */
	P eL = 0, eU = 0;                                             /* popped entry */
	uint32_t erunsLo = ~0u, erunsHi = ~0u;
	int e_i = 0, e_mm = 0, e_go = 0, e_ge = 0, e_state = 0, e_alen = 0, e_score = 0;
	unsigned long long vis_s = 0, vis_a = 0, n_pop = 0, n_push = 0, n_aln_tot = 0;

	for (;;) {
		if (!active && !done) {
			rid = next_read(b, ol, lane);
			if (rid == NONE32) done = true;
			else {
				len = b.lens[rid];
				uint32_t cntN = 0;
				const uint8_t *seq = b.reads + (size_t)rid * b.stride;
				const uint8_t *dsrc = b.dbuf + (size_t)rid * b.dstride;
				const bool has_seed = kp.seed_length && len > kp.seed_length;
				for (int k = ol; k < len; k += 8) {
					const int cf = seq[len - 1 - k];
					src[k] = (uint8_t)(cf > 3 ? 4 : 3 - cf); /* read->rc, io.c:502-504 */
					cntN += cf > 3;
					sD[k] = dsrc[k];
				}
				/* D_seed is only computed when len > seed_length (inexact_match.c:141-143); otherwise the reference
				 * reads whatever its thread's buffer holds. We define that as the calloc'd zeros. */
				for (int k = ol; k < kp.seed_length; k += 8) sDs[k] = has_seed ? dsrc[b.dseed_off + k] : (uint8_t)0x80;
				cntN = oct_add(cntN);
				for (int k = ol; k < nb; k += 8) bstate[k] = NONE32;
				n_alns = 0; mode = MODE_POP; active = true;
				h.bump = 0; h.fhead = NONE32; h.neLo = h.neHi = 0; h.best = nb; h.num_entries = 0;
				if (!((int)cntN > kp.max_diff || len == 0)) { /* inexact_match.c:260-266 */
					/* heap_push(root) inexact_match.c:281 */
					if (ol == 0) {
						cprev[0] = NONE32;
						ent[0] = make_uint4(0u, 0u, (uint32_t)(ix.length - 1), (uint32_t)((ix.length - 1) >> 32));
						ent[1] = make_uint4((uint32_t)len, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);
					}
					bstate[0] = (0u << 5) | 1u;
					h.bump = 1; h.neLo = 1; h.best = 0; h.num_entries = 1; n_push++;
				}
				best_score = kp.num_buckets; /* aln_score(max_diff+1,max_gapo+1,max_gape+1) :284 */
				max_diff = kp.max_diff; num_best = 0;
			}
		}
		if (__all(done)) break;
		if (!active) continue;

		bool finish = false, ovf = false, expanding = false, exact_iter = false, exact_done = false;
		P iL = 0, iU = 0;

		/* add_alignment (align.c:271-298) into the octet's private hit list */
		auto add_aln = [&](P L, P U, int score, int alen) {
			bool dup = false;
			if (e_go) {
				uint32_t hit = 0;
				for (int j = ol; j < n_alns; j += 8) {
					const uint4 a = myalns[j * 2];
					hit |= (a.x == (uint32_t)L && a.y == (uint32_t)((uint64_t)L >> 32) && a.z == (uint32_t)U && a.w == (uint32_t)((uint64_t)U >> 32)) ? 1u : 0u;
				}
				dup = oct_or(hit) != 0;
			}
			if (!dup) {
				if (n_alns >= (int)sc.acap) ovf = true;
				else {
					if (ol == 0) {
						myalns[n_alns * 2] = make_uint4((uint32_t)L, (uint32_t)((uint64_t)L >> 32), (uint32_t)U, (uint32_t)((uint64_t)U >> 32));
						myalns[n_alns * 2 + 1] = make_uint4((uint32_t)(score & 255) | (e_mm << 8) | (e_go << 16) | (e_ge << 24),
						                                    (uint32_t)(alen & 255), erunsLo, erunsHi);
					}
					n_alns++;
				}
			}
		};

		if (mode == MODE_POP) {
			if (h.num_entries == 0 || h.num_entries > kp.max_entries) finish = true; /* :293,299 */
			else {
				/* heap_pop :594-610 */
				const int bk = h.best;
				const uint32_t bst = bstate[bk];
				const uint32_t chunk = bst >> 5;
				const int fill = (int)(bst & 31u);
				const size_t eidx = ((size_t)chunk * 16 + fill - 1) * 2;
				const uint4 w0 = ent[eidx], w1 = ent[eidx + 1];
				if (fill == 1) {
					const uint32_t pv = cprev[chunk];
					if (ol == 0) cprev[chunk] = h.fhead; /* chunk goes to the free list */
					h.fhead = chunk;
					if (pv == NONE32) {
						bstate[bk] = NONE32;
						if (bk < 64) h.neLo &= ~(1ull << bk); else h.neHi &= ~(1ull << (bk - 64));
						h.best = h.neLo ? __ffsll((long long)h.neLo) - 1 : (h.neHi ? 64 + __ffsll((long long)h.neHi) - 1 : nb);
					} else bstate[bk] = (pv << 5) | 16u;
				} else bstate[bk] = (chunk << 5) | (uint32_t)(fill - 1);
				h.num_entries--; n_pop++;
				e_score = bk;
				eL = (P)(((uint64_t)w0.y << 32) | w0.x); eU = (P)(((uint64_t)w0.w << 32) | w0.z);
				e_i = w1.x & 255; e_mm = (w1.x >> 8) & 255; e_go = (w1.x >> 16) & 255; e_ge = (w1.x >> 24) & 255;
				e_state = w1.y & 3; e_alen = (w1.y >> 8) & 255;
				erunsLo = w1.z; erunsHi = w1.w;

				if (e_score > best_score + kp.mm_score) finish = true; /* :309 */
				else {
					const int diff_left = max_diff - e_mm - e_go - e_ge;
					const int diff_left_seed = kp.max_diff_seed - e_mm - e_go - e_ge;
					const int seed_index = e_i - (len - kp.seed_length);
					bool pruned = diff_left < 0;                                                                        /* :313 */
					if (!pruned && e_i > 0 && diff_left < (int)(sD[e_i - 1] & 127)) pruned = true;                      /* :317 */
					if (!pruned && seed_index > 0 && diff_left_seed < (int)(sDs[seed_index - 1] & 127)) pruned = true; /* :326 */
					if (!pruned) {
						if (e_i == 0) { /* hit :331-344 */
							if (n_alns == 0) {
								best_score = e_score;
								const int bd = e_mm + e_go + e_ge;
								max_diff = (bd + 1 > kp.max_diff) ? kp.max_diff : bd + 1;
							}
							if (e_score == best_score) { num_best += (int)(uint32_t)(eU - eL + 1); add_aln(eL, eU, e_score, e_alen); }
							else if (num_best > kp.max_best) finish = true;
							else add_aln(eL, eU, e_score, e_alen);
						} else if (diff_left == 0) { /* exact tail :345-375 */
							tails[0] = eL; tails[1] = eU;
							cursel = 0; curT = 1; T = 0; s = 0; r = e_i - 1;
							mode = MODE_EXACT;
						} else { expanding = true; iL = eL; iU = eU; }
					}
				}
			}
		}

		int c = 0;
		if (mode == MODE_EXACT) { /* exact_match_bounded exact_match.c:82-115 */
			c = src[r];
			if (c > 3) { curT = 0; exact_done = true; } /* :84-87 */
			else {
				if (s == curT - 1) { iL = tails[cursel * 2]; iU = tails[cursel * 2 + 1]; }
				else { const Intv<P> v = (lbase + cursel * lcap)[s]; iL = v.L; iU = v.U; }
				exact_iter = true;
			}
		}

		if (exact_iter) {
			RankReq<P> ra, rb;
			rank_issue<P>(buckets, last_row, (P)(iL - 1), ol, ra);
			rank_issue<P>(buckets, last_row, iU, ol, rb);
			vis_s += (ra.regular ? 1 : 0) + (rb.regular ? 1 : 0);
			P a0, a1, u0, u1;
			rank_finish<P, false>(ra, s_base, ol, lane, a0, a1);
			rank_finish<P, false>(rb, s_base, ol, lane, u0, u1);
			const P L0 = a0 + 1, L1 = a1 + 1;
			const uint32_t mem = member_mask(c);
			const bool ne0 = ((mem >> (2 * ol)) & 1u) && (L0 <= u0);
			const bool ne1 = ((mem >> (2 * ol + 1)) & 1u) && (L1 <= u1);
			append_children<P>(L0, u0, L1, u1, ne0, ne1, ol, lane, tails + (cursel ^ 1) * 2, lbase + (cursel ^ 1) * lcap, T, lcap, ovf);
			s++;
			if (!ovf && s >= curT) {
				cursel ^= 1; curT = T; T = 0; s = 0;
				if (curT == 0) exact_done = true; /* :114 */
				else { r--; if (r < 0) exact_done = true; }
			}
		} else if (expanding) {
			RankReq<P> ra, rb;
			rank_issue<P>(buckets, last_row, (P)(iL - 1), ol, ra);
			rank_issue<P>(buckets, last_row, iU, ol, rb);
			vis_a += (ra.regular ? 1 : 0) + (rb.regular ? 1 : 0);
			/* ---- allow_* flags :392-430 (uniform; computed while the bucket loads are in flight) ---- */
			const int diff_left = max_diff - e_mm - e_go - e_ge;
			const int diff_left_seed = kp.max_diff_seed - e_mm - e_go - e_ge;
			const int seed_index = e_i - (len - kp.seed_length);
			bool allow_diff = true, allow_indels = true, allow_mm = true, allow_open = true, allow_extend = true;
			if (e_i - 1 > 0) {
				const int d1 = sD[e_i - 1], d2 = sD[e_i - 2];
				if ((diff_left - 1) < (d2 & 127)) allow_diff = false;
				else if ((d1 & 127) == diff_left - 1 && (d2 & 127) == diff_left - 1 && (d1 & 128)) allow_mm = false;
			}
			if (seed_index - 1 > 0) {
				const int d1 = sDs[seed_index - 1], d2 = sDs[seed_index - 2];
				if ((diff_left_seed - 1) < (d2 & 127)) allow_diff = false;
				else if ((d1 & 127) == diff_left_seed - 1 && (d2 & 127) == diff_left_seed - 1 && (d1 & 128)) allow_mm = false;
			}
			const int tmp = e_go + e_ge;
			if ((e_i - 1 < kp.no_indel_length + tmp) || ((len - (e_i - 1)) < kp.no_indel_length + tmp)) allow_indels = false;
			if (e_go >= kp.max_gapo && e_ge >= kp.max_gape) allow_indels = false;
			if (e_go >= kp.max_gapo) allow_open = false;
			if (e_ge >= kp.max_gape) allow_extend = false;
			const int cr = src[e_i - 1];
			const bool gap_open = e_state == STATE_M;
			const int sc0 = e_score, scX = e_score + kp.mm_score, scG = e_score + (gap_open ? kp.gapo_score : kp.gape_score);
			const bool ins_ok = allow_diff && allow_indels && ((e_state == STATE_I && allow_extend) || (e_state == STATE_M && allow_open));
			const bool del_ok = allow_diff && allow_indels && e_state != STATE_I && (e_state == STATE_M ? allow_open : allow_extend);
			const bool mm_ok = allow_diff && allow_mm;
			const uint32_t mem = cr > 3 ? 0u : member_mask(cr);
			/* uniform parts of the child entries */
			const uint32_t alen1 = (uint32_t)((e_alen + 1) & 255);
			const uint32_t w1x_base = ((uint32_t)e_go << 16) | ((uint32_t)e_ge << 24);
			const uint32_t w1x_match = (uint32_t)((e_i - 1) & 255) | ((uint32_t)e_mm << 8) | w1x_base;
			const uint32_t w1x_mis = (uint32_t)((e_i - 1) & 255) | ((uint32_t)((e_mm + 1) & 255) << 8) | w1x_base;
			const uint32_t w1x_gap = ((uint32_t)e_mm << 8) | ((uint32_t)((e_go + (gap_open ? 1 : 0)) & 255) << 16) | ((uint32_t)((e_ge + (gap_open ? 0 : 1)) & 255) << 24);
			/* gap runs of a gap child: new run on open (start = aln_length, len 1), len+1 on extend */
			const uint64_t eruns = ((uint64_t)erunsHi << 32) | erunsLo;
			uint64_t gruns_i, gruns_d;
			if (gap_open) {
				const int sh = 16 * (e_go & 3);
				const uint64_t cleared = eruns & ~(0xFFFFull << sh);
				gruns_i = cleared | ((uint64_t)((uint32_t)e_alen | 0x100u) << sh);
				gruns_d = cleared | ((uint64_t)((uint32_t)e_alen | 0x8100u) << sh);
			} else {
				gruns_i = gruns_d = eruns + (0x100ull << (16 * ((e_go - 1) & 3)));
			}

			P a0, a1, u0, u1;
			rank_finish<P, true>(ra, s_base, ol, lane, a0, a1);
			rank_finish<P, true>(rb, s_base, ol, lane, u0, u1);
			const P L0 = a0 + 1, L1 = a1 + 1; /* inc = 1 on the L side (:382) */
			const int j0 = 2 * ol, j1 = 2 * ol + 1;
			const bool ne0 = j0 >= 1 && L0 <= u0, ne1 = L1 <= u1;
			const bool mb0 = (mem >> j0) & 1u, mb1 = (mem >> j1) & 1u;
			/* push sequence (:434-504): bit 0 insertion, bits 1..15 deletions j, bits 16+j match/mismatch j */
			const bool g0 = j0 == 0 ? ins_ok : (del_ok && ne0), g1 = del_ok && ne1;
			const bool ma0 = ne0 && mb0, ma1 = ne1 && mb1;
			const bool mi0 = mm_ok && ne0 && !mb0, mi1 = mm_ok && ne1 && !mb1;
			const uint32_t Mgm = oct_or((((uint32_t)g0 | ((uint32_t)g1 << 1)) << j0) | (((uint32_t)ma0 | ((uint32_t)ma1 << 1)) << (16 + j0)));
			const uint32_t Mx = oct_or(((uint32_t)mi0 | ((uint32_t)mi1 << 1)) << (16 + j0));
			const uint32_t Mg = Mgm & 0xFFFFu, Mm = Mgm & 0xFFFF0000u;
			/* up to three target buckets; classes that share a score share a bucket in sequence order */
			const uint32_t mA = Mm | (scX == sc0 ? Mx : 0u) | (scG == sc0 ? Mg : 0u);
			const uint32_t mB = scX != sc0 ? (Mx | (scG == scX ? Mg : 0u)) : 0u;
			const uint32_t mC = (scG != sc0 && scG != scX) ? Mg : 0u;
			Resv rA = { 0, 0, 0, 0 }, rB = { 0, 0, 0, 0 }, rC = { 0, 0, 0, 0 };
			if (mA) rA = heap_reserve(h, sc0, __popc(mA), bstate, cprev, sc.nchunks, ol, ovf);
			if (mB && !ovf) rB = heap_reserve(h, scX, __popc(mB), bstate, cprev, sc.nchunks, ol, ovf);
			if (mC && !ovf) rC = heap_reserve(h, scG, __popc(mC), bstate, cprev, sc.nchunks, ol, ovf);
			if (!ovf) {
				n_push += __popc(mA) + __popc(mB) + __popc(mC);
				/* the lane's own four possible pushes: deletion/insertion for j0, j1; match/mismatch for j0, j1 */
#pragma unroll
				for (int hh = 0; hh < 4; hh++) {
					const bool isgap = hh < 2;
					const int hsel = hh & 1;
					const bool valid = isgap ? (hsel ? g1 : g0) : (hsel ? (ma1 || mi1) : (ma0 || mi0));
					if (valid) {
						const int j = hsel ? j1 : j0;
						const bool is_mis = !isgap && (hsel ? mi1 : mi0);
						const int q = isgap ? j : 16 + j;
						const int spush = isgap ? scG : (is_mis ? scX : sc0);
						const uint32_t below = (1u << q) - 1u; /* q <= 31 */
						size_t sl;
						if (spush == sc0) sl = resv_slot(rA, __popc(mA & below));
						else if (spush == scX) sl = resv_slot(rB, __popc(mB & below));
						else sl = resv_slot(rC, __popc(mC & below));
						const bool is_ins = isgap && j == 0;
						P cL = hsel ? L1 : L0, cU = hsel ? u1 : u0;
						if (is_ins) { cL = eL; cU = eU; }
						uint32_t w1x, w1y, rl, rh;
						if (isgap) {
							w1x = w1x_gap | (uint32_t)((is_ins ? e_i - 1 : e_i) & 255);
							w1y = (is_ins ? STATE_I : STATE_D) | (alen1 << 8);
							const uint64_t gr = is_ins ? gruns_i : gruns_d;
							rl = (uint32_t)gr; rh = (uint32_t)(gr >> 32);
						} else {
							w1x = is_mis ? w1x_mis : w1x_match;
							w1y = STATE_M | (alen1 << 8);
							rl = erunsLo; rh = erunsHi;
						}
						ent[sl * 2] = make_uint4((uint32_t)cL, (uint32_t)((uint64_t)cL >> 32), (uint32_t)cU, (uint32_t)((uint64_t)cU >> 32));
						ent[sl * 2 + 1] = make_uint4(w1x, w1y, rl, rh);
					}
				}
			}
		}

		if (exact_done && !ovf) {
			mode = MODE_POP;
			if (curT != 0) { /* matches found :347-371 */
				if (n_alns == 0) {
					best_score = e_score;
					const int bd = e_mm + e_go + e_ge;
					max_diff = (bd + 1 > kp.max_diff) ? kp.max_diff : bd + 1;
				}
				bool brk = false;
				if (e_score == best_score) {
					for (int k = 0; k < curT; k++) {
						P L, U;
						if (k == curT - 1) { L = tails[cursel * 2]; U = tails[cursel * 2 + 1]; }
						else { const Intv<P> v = (lbase + cursel * lcap)[k]; L = v.L; U = v.U; }
						num_best += (int)(uint32_t)(U - L + 1);
					}
				} else if (num_best > kp.max_best) brk = true;
				if (brk) finish = true;
				else {
					const int alen2 = (e_alen + e_i) & 255; /* :365 */
					for (int k = 0; k < curT && !ovf; k++) {
						P L, U;
						if (k == curT - 1) { L = tails[cursel * 2]; U = tails[cursel * 2 + 1]; }
						else { const Intv<P> v = (lbase + cursel * lcap)[k]; L = v.L; U = v.U; }
						add_aln(L, U, e_score, alen2);
					}
				}
			}
		}

		if (ovf) finish = true;
		if (finish) {
			unsigned long long off = 0;
			bool outovf = false;
			if (!ovf && n_alns > 0) {
				if (ol == 0) off = atomicAdd(out.count, (unsigned long long)n_alns);
				off = oct_bcast((uint64_t)off, lane & ~7);
				if (off + (unsigned long long)n_alns > out.cap) outovf = true;
				else for (int t = ol; t < n_alns * 2; t += 8) out.alns[off * 2 + t] = myalns[t];
			}
			if (ol == 0) {
				out.off[rid] = off;
				out.n[rid] = (ovf || outovf) ? 0u : (uint32_t)n_alns;
				b.status[rid] = ovf ? ST_SCRATCH_OVF : (outovf ? ST_OUT_OVF : ST_OK);
			}
			if (!ovf && !outovf) n_aln_tot += n_alns;
			active = false;
		}
	}
	if (ol == 0) {
		if (vis_s) atomicAdd(&stats[STAT_VIS_SINGLE], vis_s);
		if (vis_a) atomicAdd(&stats[STAT_VIS_ALPHA], vis_a);
		if (n_pop) atomicAdd(&stats[STAT_POPS], n_pop);
		if (n_push) atomicAdd(&stats[STAT_PUSHES], n_push);
		if (n_aln_tot) atomicAdd(&stats[STAT_ALNS], n_aln_tot);
	}
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
	buckets[blk * 8 + sl] = o;
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
			acc += (unsigned long long)v0 + 3ull * v1;
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

/* SA[row] by the invPsi walk (bwt.c:311-329): one octet per row */
__global__ __launch_bounds__(BWB_BLOCK) void k_locate(DevIndex ix, const uint64_t *SA, uint64_t sa0_index, const uint64_t *rows, uint64_t n, uint64_t *out) {
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
			/* B(i), bwt.c:337-345: bit (i&31) of the planes held by lane 4 + ((i&127)>>5) */
			const uint4 cq = ra.regular ? ra.q : ix.buckets[(i >> 7) * 8 + ol]; /* i == length-1 is not ranked via its bucket */
			const int off = (int)(i & 127), srcl = (lane & ~7) + 4 + (off >> 5), bit = off & 31;
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
		if (ol == 0) out[q] = (SA[i >> 5] + j) % ix.length;
	}
}
