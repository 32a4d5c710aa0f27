/*
 * bwb_device.h - device-side FM-index layout and the octet-cooperative rank primitive (gfx950).
 *
 * Execution model: one read is owned by an OCTET = 8 adjacent lanes of a 64-wide wavefront (8 reads
 * per wave).  A rank-block visit is ONE coalesced 128-byte load: lane k of the octet loads bytes
 * [16k, 16k+16) of the bucket with a single global_load_dwordx4.
 *
 * Bucket (128 B, 128-byte aligned) for the 64 BWT positions [64b, 64b+63]  -- replaces the reference's
 * separate O row (128 B per 128 positions, bwt.c:288) + packed BWT words (io.c:590-609):
 *   slice s=0..3 : uint32 {cnt[2s], cnt[2s+1], cnt[2s+8], cnt[2s+9]}
 *                  cnt[c] = #c in BWT[superblock_start .. 64b-1]  (EXCLUSIVE of this bucket, sentinel
 *                  row excluded as in compute_O bwt.c:284).
 *   slice 4+w    : uint32 {p0,p1,p2,p3} bit-planes of characters [32w, 32w+32), w = 0, 1: bit j of p_k is
 *                  bit k of the 4-bit code at bucket offset 32w+j.
 *   slice 6      : #c among the first 32 characters as bytes: component s = {mid[2s], mid[2s+1], mid[2s+8],
 *                  mid[2s+9]} (the order of the count slices), so that a rank of a position in the second
 *                  sub-block is counts + mid + ONE masked popcount pass (the lane kernels, bwb_lane.h).
 *   slice 7      : .x = code of the first character of the enclosing 128-character block (what O_alphabet
 *                  corrects by, bwt.c:780).
 * 2 bytes of index per BWT character.  A 128-character bucket (1 byte per character) costs a rank four masked
 * passes per side: measured 8-10 % slower in the alignment kernels (profiles/r3_bkt64_ab.txt); that layout and
 * its lane path are kept as bwbble_amd/tools_exp/bkt128_lane_path.patch.
 * Absolute counts need > 32 bits on GRCh37-scale texts, so a superblock (2^24 blocks = 2^31 chars)
 * base table base[sb][c] = C[c] + #c before the superblock lives in LDS (<= 10 rows).  Blocks below are the
 * reference's 128-character O rows (DevIndex::nblk); a superblock is 2^(BWB_SB_SHIFT+1) buckets.
 *
 * Positions are a template parameter P: uint32_t when the BWT has < 2^32 - 1 rows (everything up
 * to ~2 G forward characters), uint64_t otherwise (GRCh37 + 1000G).  "-1" is ~P(0).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BWB_SB_SHIFT
#define BWB_SB_SHIFT 24                 /* blocks per superblock = 2^24 (the test build `make testlib` uses 13, so that a 6 M-row index
                                           spans several superblocks like a 6.85 G-row one does) */
#endif
#define BWB_NSB_MAX 8                   /* up to 2^34 BWT characters */
#define BWB_ROW_NEG (BWB_NSB_MAX)       /* base row for position -1      : C[j]   (bwt.c:393-410) */
#define BWB_ROW_END (BWB_NSB_MAX + 1)   /* base row for position length-1: C[j+1] (bwt.c:375-392) */
#define BWB_BASE_ROWS (BWB_NSB_MAX + 2)

#define BKT_SHIFT 6                     /* characters per bucket = 64 */
#define BKT_MASK ((1 << BKT_SHIFT) - 1)
#define BKT_SB_SHIFT (BWB_SB_SHIFT + 7 - BKT_SHIFT) /* buckets per superblock: 2^31 characters either way */

struct DevIndex {
	const uint4 *buckets;               /* 2 * nblk * 8 slices: one 128-byte bucket per 64 BWT characters */
	uint64_t length;                    /* n + 1 (bwt_t.length) */
	uint64_t nblk;                      /* 128-character blocks (the reference's O rows) */
	uint64_t base[BWB_BASE_ROWS][16];
};

/* ---- intra-octet data movement on the DPP path (no LDS traffic) ------------------------------ */
#define DPP_QUAD_XOR1 0xB1      /* quad_perm:[1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E      /* quad_perm:[2,3,0,1] */
#define DPP_HALF_MIRROR 0x141   /* row_half_mirror: lane i <- lane 7-i of its 8-lane half row */
#define DPP_ROW_SHR4 0x114      /* row_shr:4 */

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
/* all-reduce over the 8 lanes of an octet.  After the two quad steps every lane of a quad holds the
 * quad's value, so the mirrored half row delivers "the other quad" to each lane. */
__device__ __forceinline__ uint32_t oct_or(uint32_t v) {
	v |= dpp<DPP_QUAD_XOR1>(v); v |= dpp<DPP_QUAD_XOR2>(v); v |= dpp<DPP_HALF_MIRROR>(v);
	return v;
}
__device__ __forceinline__ uint32_t oct_add(uint32_t v) {
	v += dpp<DPP_QUAD_XOR1>(v); v += dpp<DPP_QUAD_XOR2>(v); v += dpp<DPP_HALF_MIRROR>(v);
	return v;
}
/* lanes 4..7 of every octet receive the value of lane-4 (lanes 0..3 receive garbage/0) */
__device__ __forceinline__ uint32_t oct_shr4(uint32_t v) { return dpp<DPP_ROW_SHR4>(v); }

__device__ __forceinline__ uint32_t oct_bcast(uint32_t v, int lane_in_wave) { return (uint32_t)__shfl((int)v, lane_in_wave); }
__device__ __forceinline__ uint64_t oct_bcast(uint64_t v, int lane_in_wave) {
	uint32_t lo = __shfl((uint32_t)v, lane_in_wave), hi = __shfl((uint32_t)(v >> 32), lane_in_wave);
	return ((uint64_t)hi << 32) | lo;
}

/* One side of a visit pair, before the data arrives */
template <typename P> struct RankReq {
	P pos;
	bool regular;                       /* false for pos == -1 / length-1: no memory touched */
	int row;                            /* base-table row */
	uint4 q;                            /* this lane's 16-byte slice */
};

template <typename P>
__device__ __forceinline__ void rank_issue(const uint4 *__restrict__ buckets, P last_row, P pos, int ol, RankReq<P> &r) {
	r.pos = pos;
	const bool neg = (pos == (P)~(P)0), end = (pos == last_row);
	r.regular = !(neg || end);
	const P blk = pos >> BKT_SHIFT;
	r.row = neg ? BWB_ROW_NEG : (end ? BWB_ROW_END : (int)((uint64_t)blk >> BKT_SB_SHIFT));
	r.q = make_uint4(0, 0, 0, 0);
	if (r.regular) r.q = buckets[(size_t)blk * 8 + ol];
}

/*
 * Finishes a visit: v0/v1 = C[j] + Occ(j, pos) for this lane's codes j = 2*ol and 2*ol+1.
 * QUIRK reproduces O_alphabet (bwt.c:423-437): codes 5, 9, 11, 13 get
 * C[j] - [first char of the block == j] instead of their count.  s_base is the LDS copy of
 * DevIndex::base (as P).
 */
template <typename P, bool QUIRK>
__device__ __forceinline__ void rank_finish(const RankReq<P> &r, const P *s_base, int ol, int lane, P &v0, P &v1, uint32_t *first_out = nullptr) {
	const uint4 q = r.q;
	/* position mask of this lane's 32-character sub-block (lanes 4 and 5 hold planes; 0..3 counts, 6 the mid counts, 7 the first character: empty mask) */
	const int off = (int)(r.pos & BKT_MASK);
	const int nvalid = off + 1 - 32 * (ol - 4);
	const uint32_t m = (ol < 4 || ol > 5 || nvalid <= 0 || !r.regular) ? 0u : (nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u));
	const uint32_t p0 = q.x, p1 = q.y, p2 = q.z, p3 = q.w;
	const uint32_t n0 = ~p0, n1 = ~p1;
	const uint32_t a0 = n0 & n1, a1 = p0 & n1, a2 = n0 & p1, a3 = p0 & p1;                 /* code & 3  */
	const uint32_t m2 = m & ~p2, m2p = m & p2;
	const uint32_t b0 = m2 & ~p3, b1 = m2p & ~p3, b2 = m2 & p3, b3 = m2p & p3;             /* code >> 2 */
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
	const uint32_t cz = oct_shr4(q.z), cw = oct_shr4(q.w);
	uint32_t c0 = ol < 4 ? q.x : cz, c1 = ol < 4 ? q.y : cw;
	if (!r.regular) { c0 = 0; c1 = 0; }
	const P *brow = s_base + r.row * 16 + 2 * ol;
	v0 = brow[0] + c0 + pop0;
	v1 = brow[1] + c1 + pop1;
	if (first_out) /* first character of the enclosing 128-character block: slice 7 */
		*first_out = oct_bcast(q.x, (lane & ~7) + 7);
	if (QUIRK) {
		/* first character of the enclosing 128-character block (bwt.c:780): slice 7 */
		const uint32_t first = oct_bcast(q.x, (lane & ~7) + 7);
		const int j1 = 2 * ol + 1; /* only odd codes 5, 9, 11, 13 (lanes 2, 4, 5, 6) */
		if (r.regular && (j1 == 5 || j1 == 9 || j1 == 11 || j1 == 13))
			v1 = s_base[BWB_ROW_NEG * 16 + j1] - (first == (uint32_t)j1 ? (P)1 : (P)0);
	}
}
