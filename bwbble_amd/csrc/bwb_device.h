/*
 * bwb_device.h - device-side FM-index layout and the octet-cooperative rank primitive (gfx950).
 *
 * Execution model: one read is owned by an OCTET = 8 adjacent lanes of a 64-wide wavefront (8 reads
 * per wave).  A rank-block visit is ONE coalesced 128-byte load: lane k of the octet loads bytes
 * [16k, 16k+16) of the bucket with a single global_load_dwordx4.
 *
 * Bucket (128 B, 128-byte aligned) for BWT positions [128b, 128b+127]  -- replaces the reference's
 * separate O row (128 B, bwt.c:288) + packed BWT words (64 B, io.c:590-609):
 *   slice s=0..3 : uint32 {cnt[2s], cnt[2s+1], cnt[2s+8], cnt[2s+9]}
 *                  cnt[c] = #c in BWT[superblock_start .. 128b-1]  (EXCLUSIVE of this block, sentinel
 *                  row excluded as in compute_O bwt.c:284).
 *   slice 4+w    : uint32 {p0,p1,p2,p3} bit-planes of characters [32w, 32w+32): bit j of p_k is bit k
 *                  of the 4-bit code at block offset 32w+j.
 * Absolute counts need > 32 bits on GRCh37-scale texts, so a superblock (2^24 blocks = 2^31 chars)
 * base table base[sb][c] = C[c] + #c before the superblock lives in LDS (<= 10 rows x 128 B).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BWB_SB_SHIFT 24                 /* blocks per superblock = 2^24 */
#define BWB_NSB_MAX 8                   /* up to 2^34 BWT characters */
#define BWB_ROW_NEG (BWB_NSB_MAX)       /* base row for position -1      : C[j]   (bwt.c:393-410) */
#define BWB_ROW_END (BWB_NSB_MAX + 1)   /* base row for position length-1: C[j+1] (bwt.c:375-392) */
#define BWB_BASE_ROWS (BWB_NSB_MAX + 2)

struct DevIndex {
	const uint4 *buckets;               /* nblk * 8 slices */
	uint64_t length;                    /* n + 1 (bwt_t.length) */
	uint64_t nblk;
	uint64_t base[BWB_BASE_ROWS][16];
};

__device__ __forceinline__ uint32_t oct_or(uint32_t v) {
	v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4);
	return v;
}
__device__ __forceinline__ uint32_t oct_add(uint32_t v) {
	v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
	return v;
}
__device__ __forceinline__ uint64_t oct_bcast64(uint64_t v, int lane_in_wave) {
	uint32_t lo = __shfl((uint32_t)v, lane_in_wave), hi = __shfl((uint32_t)(v >> 32), lane_in_wave);
	return ((uint64_t)hi << 32) | lo;
}

/* One side of a visit pair, before the data arrives */
struct RankReq {
	uint64_t pos;
	bool regular;                       /* false for pos == -1 / length-1: no memory touched */
	int row;                            /* base-table row */
	uint4 q;                            /* this lane's 16-byte slice */
};

__device__ __forceinline__ void rank_issue(const DevIndex &ix, uint64_t pos, int ol, RankReq &r) {
	r.pos = pos;
	const bool neg = (pos == ~0ull), end = (pos == ix.length - 1);
	r.regular = !(neg || end);
	const uint64_t blk = pos >> 7;
	r.row = neg ? BWB_ROW_NEG : (end ? BWB_ROW_END : (int)(blk >> BWB_SB_SHIFT));
	r.q = make_uint4(0, 0, 0, 0);
	if (r.regular) r.q = ix.buckets[blk * 8 + ol];
}

/*
 * Finishes a visit: v0/v1 = C[j] + Occ(j, pos) for this lane's codes j = 2*ol and 2*ol+1.
 * quirk != 0 reproduces O_alphabet (bwt.c:423-437): codes 5, 9, 11, 13 get
 * C[j] - [first char of the block == j] instead of their count.  s_base is the LDS copy of
 * DevIndex::base.
 */
__device__ __forceinline__ void rank_finish(const RankReq &r, const uint64_t *s_base, int ol, int lane, bool quirk,
                                            uint64_t &v0, uint64_t &v1) {
	const uint4 q = r.q;
	/* position mask of this lane's 32-character sub-block (lanes 0..3 hold counts: empty mask) */
	const int off = (int)(r.pos & 127);
	const int nvalid = off + 1 - 32 * (ol - 4);
	uint32_t m = (ol < 4 || nvalid <= 0 || !r.regular) ? 0u : (nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
	const uint32_t p0 = q.x, p1 = q.y, p2 = q.z, p3 = q.w;
	const uint32_t n0 = ~p0, n1 = ~p1, n2 = ~p2, n3 = ~p3;
	const uint32_t a0 = n0 & n1, a1 = p0 & n1, a2 = n0 & p1, a3 = p0 & p1;           /* code & 3  */
	const uint32_t b0 = n2 & n3 & m, b1 = p2 & n3 & m, b2 = n2 & p3 & m, b3 = p2 & p3 & m; /* code >> 2 */
	uint32_t pc0 = __popc(a0 & b0) | (__popc(a1 & b0) << 8) | (__popc(a2 & b0) << 16) | (__popc(a3 & b0) << 24);
	uint32_t pc1 = __popc(a0 & b1) | (__popc(a1 & b1) << 8) | (__popc(a2 & b1) << 16) | (__popc(a3 & b1) << 24);
	uint32_t pc2 = __popc(a0 & b2) | (__popc(a1 & b2) << 8) | (__popc(a2 & b2) << 16) | (__popc(a3 & b2) << 24);
	uint32_t pc3 = __popc(a0 & b3) | (__popc(a1 & b3) << 8) | (__popc(a2 & b3) << 16) | (__popc(a3 & b3) << 24);
	/* byte-wise all-reduce over the octet: each byte <= 128, no carries */
	pc0 = oct_add(pc0); pc1 = oct_add(pc1); pc2 = oct_add(pc2); pc3 = oct_add(pc3);
	const int hs = ol >> 1;
	const uint32_t pd = hs == 0 ? pc0 : (hs == 1 ? pc1 : (hs == 2 ? pc2 : pc3));
	const int sh = (ol & 1) * 16;
	const uint32_t pop0 = (pd >> sh) & 0xFF, pop1 = (pd >> (sh + 8)) & 0xFF;
	/* counts: lanes 0..3 own (x,y); lanes 4..7 take (z,w) of lane ol-4 */
	const uint32_t cz = __shfl_up(q.z, 4, 8), cw = __shfl_up(q.w, 4, 8);
	/* first character of the block (for the bwt.c:780 quirk): bit 0 of the four planes held by lane 4 */
	const uint32_t first = __shfl((q.x & 1u) | ((q.y & 1u) << 1) | ((q.z & 1u) << 2) | ((q.w & 1u) << 3), (lane & ~7) + 4);
	uint32_t c0 = ol < 4 ? q.x : cz, c1 = ol < 4 ? q.y : cw;
	if (!r.regular) { c0 = 0; c1 = 0; }
	const uint64_t *brow = s_base + r.row * 16 + 2 * ol;
	v0 = brow[0] + c0 + pop0;
	v1 = brow[1] + c1 + pop1;
	if (quirk && r.regular) {
		/* only odd codes 5, 9, 11, 13 (lanes 2, 4, 5, 6) */
		const int j1 = 2 * ol + 1;
		if (j1 == 5 || j1 == 9 || j1 == 11 || j1 == 13) v1 = s_base[BWB_ROW_NEG * 16 + j1] - (first == (uint32_t)j1 ? 1u : 0u);
	}
}
