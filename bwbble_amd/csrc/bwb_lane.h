/*
 * bwb_lane.h - the alignment kernels, ONE READ PER LANE (gfx950).  Included by bwb_hip.hip.
 *
 * k_calc_d : calculate_d for the full read and for the seed (inexact_match.c:171-254, called at
 *            inexact_match.c:140-143) -> one byte per position (num_diff clamped to 127, bit 7 =
 *            "sa_intv_width equals the previous position's", the only way the width is ever used,
 *            inexact_match.c:402-403,411-412).
 * k_search : inexact_match (inexact_match.c:256-506) with exact_match_bounded (exact_match.c:66-119)
 *            as a mode of the same per-lane loop: every loop iteration of every lane is one
 *            rank visit pair (positions L-1 and U of one SA interval).
 *
 * Why one read per lane: all control logic (pop, prune checks, allow flags, heap bookkeeping) serves 64 reads per wave instruction;
 * the first version of this path gave a read to an octet of lanes and was VALU-bound at 8 reads per wave instruction.  A rank visit
 * needs one 128-byte bucket per lane: the wave gathers the 64 (+ the second buckets of the pairs that straddle two) cooperatively
 * into LDS - 8 lines of 128 contiguous bytes per instruction instead of 64 different ones, which at GRCh37 scale is 50 against
 * 10.9 G buckets/s (address translation, not HBM, binds the per-lane shape; tools_exp/gather_bench.hip) - and every lane ranks its
 * own rows straight from LDS (wave_children).  Lanes pull reads from a global cursor (work stealing).
 *
 * Per-lane memory (global, private to the lane while it owns a read):
 *   heap    : chains of 64-slot chunks; a lane owns a private run of `keep` chunks, what a read takes beyond them comes
 *             from the block's lock-free stack of recycled chunks or the shared pool (atomic bump allocation) and goes
 *             to that stack when the read ends.  slot 0 = header.
 *   bstate  : [slot][bucket] u32 = chunk<<6 | fill; the bucket being popped is cached in registers.
 *   lists   : two SA-interval lists (cur/next) with the open tail in registers.
 *   hits    : the read's alignments (needed for the gapped-duplicate check, align.c:273-280).
 * Per-read (not per-lane) data written by k_calc_d for k_search: for every read position i = 1..len the two
 * bytes {D[i-1], D[i-2]} and {Dseed[si-1], Dseed[si-2]} (si = i - (len - seed_length)), plus the read's N count.
 */
#pragma once
#include "bwb_device.h"
#include "bwb_kernels.h"

#define LANE_BLOCK 256
/* Both alignment kernels are built for THREE waves per SIMD (three 256-thread blocks per CU): at most 168 registers per lane and
 * 52.6 KB of LDS per block.  Round 2 ran two (256 registers, 66 KB): the waves spent half their time waiting on memory with the
 * SIMDs' issue slots a third used; the third wave is what the round-3 register and LDS diet (rank from LDS, children in relative
 * form, list selector, window mask of the non-empty buckets, statistics straight to memory) is for. */
#ifndef LANE_WAVES_PER_SIMD
#define LANE_WAVES_PER_SIMD 3
#endif
/* lane-private LDS columns: explicit LDS address space, so they compile to ds_read/ds_write (a generic or volatile
 * pointer here turns every access into a flat_* instruction with 64-bit addresses and full waits) */
template <typename P> using Lds = __attribute__((address_space(3))) P *;
/* wave votes on a BOOLEAN: the lane mask itself.  HIP's __ballot / __any / __all take an int, and the compiler turns the bool into 0 / 1 in a
 * vector register and compares it with zero again - two vector instructions and a hazard nop per vote, fourteen votes per iteration of
 * kl_search (tools/bbprof.py: 22 vector instructions per wave iteration under amd_warp_functions.h / amd_device_functions.h) */
__device__ __forceinline__ unsigned long long wballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool wany(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
__device__ __forceinline__ bool wall(bool b) { return __builtin_amdgcn_ballot_w64(!b) == 0ull; } /* (over the lanes that execute it: inactive lanes vote 0) */
#define CHUNK_SLOTS 64          /* slot 0 is the header: 63 entries per chunk */
#define POOL_REGIONS 8
#define BSTATE_ROW_MIN 128      /* bucket states per lane (LaneScratch::brow): at least this many, else the score range rounded up to 64 (at most 1024 buckets, bwb_hip.hip check_params) */
#define PRECALC_LEN 12            /* PRECALC_INTERVAL_LENGTH align.h:31 */
#define ADMIT_CHUNKS 1024        /* free chunks a block wants to see per read it starts once the pool runs low */
#define SAVE_U4 16               /* uint4 per lane in the save area */
#define ALN_U4 3                 /* uint4 per hit record: bwb_aln is 48 bytes (include/bwbble_hip.h) */

struct LaneScratch {
	uint4 *pool;                /* chunk pool in POOL_REGIONS regions (block b uses region b % n_regions: its XCD's when all 8 are in use):
	                               chunk c of a region at region base + c * CHUNK_SLOTS * (WIDE ? 2 : 1) */
	size_t region_u4;           /* uint4 per region */
	uint32_t n_regions;         /* regions this launch uses (<= POOL_REGIONS) */
	unsigned int *pool_bump;    /* [POOL_REGIONS * 16] next never-used chunk of every region, 64 bytes apart */
	uint32_t pool_cap;          /* chunks in a region */
	uint32_t keep;              /* chunks a lane keeps for itself across reads */
	uint32_t *bstate;           /* [nslots][128]: a lane's bucket states are contiguous (an expansion touches scores that are
	                               a few apart: one or two lines instead of one line per bucket) */
	void *lists;                /* [nslots][2*lcap] Intv<P>: kl_search (multi-interval exact tails) */
	void *lists_d;              /* the same for kl_calc_d, which runs between two slices of kl_search while parked reads keep theirs */
	uint4 *save;                /* [nslots][SAVE_U4]: a lane's read, parked at the end of a slice (word 0 bit 0 = occupied) */
	uint32_t *blocksave;        /* [blocks][4]: a block's recycle stack {~head lo, ~head hi, chunks, -} across slices */
	uint4 *alns;                /* [nslots][acap*2] */
	uint32_t nslots, lcap, acap;
	uint32_t brow;              /* bucket states per lane */
};

/* ---- per-lane rank from the wave's LDS staging area ------------------------------------------------------------------------
 *
 * LDS of a wave (WAVE_LDS_BYTES = 12.5 KB; three blocks of four waves fit a CU's 160 KB):
 *   [0, 8 KB)        64 rows of 128 bytes, one per lane.  During the gather: the bucket of position L-1 of the lane ("L row").
 *                    After the rank: the lane's CHILDREN in relative form - 16 + 16 words: relL[j], relU[j] = #j in
 *                    BWT[superblock start .. L-1] resp. [.. U] (the same 32-bit counts a bucket stores, plus the block's share),
 *                    so that child j = [base[rowL][j] + relL[j] + 1, base[rowU][j] + relU[j]] (kid_get).  A visit needs the
 *                    interval of one or two children (a match, now and then a mismatch; deletions travel as one group entry),
 *                    so the 15 x 2 positions are no longer materialised: round 2 wrote 30 of them per visit.
 *   [8 KB, 11 KB)    NU_MAX = 24 rows: the bucket of position U of the lanes whose U falls into ANOTHER bucket, compacted (22 % of the
 *                    pairs at GRCh37 scale: 14 of 64 lanes on average; when more than NU_MAX lanes of a wave need one - the first
 *                    positions of calculate_d, whose intervals span the whole index - the rest are fetched in a further round).
 *                    Round 3 had 32 rows: one more load instruction in every iteration for rows that are almost never used
 *                    (+1.6 % reads/s with 24, profiles/r4_ab_steps.txt session 11)
 *   [11 KB, 11.5 KB) the exchange array of the gather
 * A rank is computed straight from the LDS rows, one 32-character sub-block at a time (four plane words in registers, not the
 * 2 x 32 words of both buckets as in round 2: the two kernels' register budgets are what allows three waves per SIMD). */
#ifndef NU_MAX
#define NU_MAX 24 /* compacted U rows per round (a multiple of 8, at most 32: the exchange array holds four per column) */
#endif
/* (the number of U rows is a template parameter of the gather: kl_search and kl_calc_d choose their own, CALCD_NU below) */
#define WAVE_XCH_OFF_NU(nu) ((64 * 8 + (nu) * 8) * 16)
#define WAVE_LDS_BYTES_NU(nu) (WAVE_XCH_OFF_NU(nu) + 512)
#define WAVE_XCH_OFF WAVE_XCH_OFF_NU(NU_MAX)
#define WAVE_LDS_BYTES WAVE_LDS_BYTES_NU(NU_MAX)
/* block-level LDS in front of the waves' areas: the base table, then one all-zero 128-byte row that stands in for the bucket of a
 * position that needs none (-1, length-1, an idle lane): counts 0, no characters */
/* (after the base table its superblock rows a second time, as O_alphabet sees them - load_base2 -; kl_calc_d only reads the first.  LDS is
 * what three blocks per CU hang on: this second table, when the waves' areas still had 32 U rows, left room for two blocks only - 13 %
 * fewer reads/s, while the occupancy query still answered three: profiles/r4_ab_steps.txt sessions 9-12) */
#define LDS_ZERO_OFF ((BWB_BASE_ROWS + BWB_NSB_MAX) * 16 * 8)
#define LDS_WAVES_OFF (LDS_ZERO_OFF + 128)
/* Three blocks per CU: the CU's 160 KB are handed out in units of 1 280 bytes (measured, bwbble_amd/tools_exp/lds_probe.hip ->
 * profiles/r4_lds_probe.txt: 53 760 bytes per block give three resident blocks, 53 888 give two while the occupancy query still says
 * three), static LDS of the kernel included (kl_search: 160 bytes). */
#define LDS_CU_BYTES 163840
#define LDS_GRANULE 1280
static_assert(3 * (((LDS_WAVES_OFF + 4 * WAVE_LDS_BYTES + 128 + 256 + LDS_GRANULE - 1) / LDS_GRANULE) * LDS_GRANULE) <= LDS_CU_BYTES, "three blocks of four waves per CU no longer fit the LDS");

/* kl_calc_d has its own LDS map: the first base table only (exact counts), the zero row, then its waves' areas with CALCD_NU compacted U rows.
 * CALCD_WAVES_PER_SIMD = 4 needs at most 40 960 bytes per block (32 granules): 1 408 + 4 x 9 728 = 40 320 with 8 U rows. */
#ifndef CALCD_NU
#define CALCD_NU NU_MAX
#endif
#ifndef CALCD_WAVES_PER_SIMD
#define CALCD_WAVES_PER_SIMD LANE_WAVES_PER_SIMD
#endif
#define CALCD_ZERO_OFF (BWB_BASE_ROWS * 16 * 8)
#define CALCD_WAVES_OFF (CALCD_ZERO_OFF + 128)
#define LDS_ALIGN_SLACK 128 /* the kernels align their dynamic LDS to 128 bytes themselves */
#define CALCD_LDS_BYTES (CALCD_WAVES_OFF + (LANE_BLOCK / 64) * WAVE_LDS_BYTES_NU(CALCD_NU) + LDS_ALIGN_SLACK)
static_assert(CALCD_WAVES_PER_SIMD * (((CALCD_LDS_BYTES + LDS_GRANULE - 1) / LDS_GRANULE) * LDS_GRANULE) <= LDS_CU_BYTES, "kl_calc_d: its blocks per CU no longer fit the LDS");
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); /* (a plain vector: HIP's uint4 class has no LDS-address-space operators) */
typedef __attribute__((address_space(3))) unsigned char *LdsBytes;

/* cache policy of the bucket loads (aux operand: 1 = sc0, 2 = nt, 16 = sc1).  A non-temporal policy, meant to keep the per-lane
 * metadata in L2, measured 2 % slower at 884 M rows (round 2): left at the default. */
#ifndef BWB_GATHER_AUX
#define BWB_GATHER_AUX 0
#endif
#define QUIRK_CODES 0x2A20u /* codes 5, 9, 11, 13: not counted by O_alphabet (bwt.c:718-726) */

template <typename P> __device__ __forceinline__ bool pi_regular(P last_row, P pos) { return !(pos == (P)~(P)0 || pos == last_row); }

/* what a lane knows about its pair of positions */
template <typename P> struct PairInfo {
	int offL, offU;        /* position & 127 */
	int rowL, rowU;        /* base-table rows */
	bool regL, regU;       /* false for -1 / length-1 (no bucket: the base row is the answer) */
	bool same;             /* both regular, one bucket */
	uint32_t blkL, blkU;   /* bucket to fetch for either side, NONE32 = none (block numbers fit 32 bits: 2^34 characters / 128) */
	uint32_t ku;           /* this lane's rank among the lanes of the wave that fetch a U row (NONE32: it fetches none) */
	int nU;                /* how many lanes of the wave do */
};
template <typename P>
__device__ __forceinline__ void pair_setup(P last_row, bool need, P pL, P pU, int lane, PairInfo<P> &pi) {
	const bool negL = (pL == (P)~(P)0), endL = (pL == last_row), negU = (pU == (P)~(P)0), endU = (pU == last_row);
	pi.regL = !(negL || endL); pi.regU = !(negU || endU);
	pi.offL = (int)(pL & BKT_MASK); pi.offU = (int)(pU & BKT_MASK);
	const P bL = pL >> BKT_SHIFT, bU = pU >> BKT_SHIFT;
	pi.rowL = negL ? BWB_ROW_NEG : (endL ? BWB_ROW_END : (int)((uint64_t)bL >> BKT_SB_SHIFT));
	pi.rowU = negU ? BWB_ROW_NEG : (endU ? BWB_ROW_END : (int)((uint64_t)bU >> BKT_SB_SHIFT));
	pi.same = need && pi.regL && pi.regU && bL == bU;
	const bool wantL = need && pi.regL, wantU = need && pi.regU && !pi.same;
	pi.blkL = wantL ? (uint32_t)bL : NONE32; pi.blkU = wantU ? (uint32_t)bU : NONE32;
	const unsigned long long maskU = wballot(wantU);
	pi.nU = __popcll(maskU);
	pi.ku = wantU ? (uint32_t)__popcll(maskU & ((1ull << lane) - 1ull)) : NONE32;
}

/* Cooperative gather: one 128-byte bucket per lane and side, loaded by the whole wave.  Instruction r fetches the buckets of the
 * owners 8r .. 8r+7: lane l loads slice (l & 7) of the bucket of owner 8r + (l >> 3), so a wave instruction touches 8 lines of
 * 128 contiguous bytes instead of 64 different ones (measured on a 7 GiB table: 50 G buckets/s this way against 10.9 G/s when every
 * lane loads its own bucket with 8 x dwordx4 - that shape is bound by address translation, 64 pages per instruction).  The loads
 * are global_load_lds_dwordx4: memory -> LDS without a register in between, all in flight together; such a load writes LDS at
 * base + 16 * lane, so to keep the owners' 128-bit reads off each other's banks the slices of row o are permuted on the SOURCE side: slot p
 * of row o holds slice p ^ rot(o), rot(o) = (o >> 1) & 7 (round 5: an XOR, where rounds 3-4 rotated by an addition - with rows that are
 * 128-byte aligned the address of slice s of a row is then (row + 16 rot) ^ 16 s, ONE instruction instead of add / and / shift-add: RowRef).  Which bucket an owner wants travels through the wave's exchange array (written transposed, so that a lane reads
 * the 8 owners of its column with 128-bit reads; round 2 used 16 ds_bpermute).
 * Round `first` == 0 fetches the L rows and the U rows of the compacted owners [0, NU_MAX); a later round (first = NU_MAX, ...)
 * only the U rows of the owners [first, first + NU_MAX).  Called by EVERY lane of the wave. */
template <typename P, int NU = NU_MAX>
__device__ __forceinline__ void wave_gather(const uint4 *__restrict__ buckets, const PairInfo<P> &pi, int first, Lds<u32x4> stage, int lane) {
	const int sub = lane >> 3, p = lane & 7;
	Lds<uint32_t> xch = (Lds<uint32_t>)((LdsBytes)stage + WAVE_XCH_OFF_NU(NU)); /* [0,64): L owners, [64,128): U owners, both transposed: owner o at (o & 7) * 8 + (o >> 3) */
	const bool mineU = pi.ku != NONE32 && (int)pi.ku >= first && (int)pi.ku < first + NU;
	const uint32_t k = pi.ku - (uint32_t)first;
	/* (the exchange array carries bucket << 3, the bucket's index in 16-byte slices - one add and one shift-add make the address; a bucket
	 * number has 28 bits.  An owner with nothing to fetch publishes bucket 0: every lane then loads in every instruction, with no exec mask
	 * to narrow and restore - compare, s_and_saveexec, branch and s_or were four of the nine instructions of a load, eleven times per
	 * iteration (tools/bbprof.py).  The idle slots' 128 bytes are one hot line of the L1; they land in rows that nobody reads: the row of a
	 * lane without a request, or a U row beyond the wave's count.) */
	if (first == 0) xch[(lane & 7) * 8 + (lane >> 3)] = pi.blkL == NONE32 ? 0u : pi.blkL << 3;
	xch[64 + lane] = 0u;
	if (mineU) xch[64 + (k & 7) * 8 + (k >> 3)] = pi.blkU << 3;
	u32x4 a0 = { 0u, 0u, 0u, 0u }, a1 = a0;
	if (first == 0) { a0 = ((Lds<u32x4>)xch)[sub * 2]; a1 = ((Lds<u32x4>)xch)[sub * 2 + 1]; }
	const u32x4 b0 = ((Lds<u32x4>)xch)[16 + sub * 2];
	__builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the owners are in registers, and whatever was read from the rows before (the children of
	                                       the previous iteration, the previous round's U rows) is too */
	/* the LDS destination of a load (M0) from a scalar copy of the wave's LDS base: no vector add and readfirstlane per load */
	uint32_t sv = (uint32_t)(uintptr_t)stage;
	asm volatile("" : "+v"(sv)); /* (not loop-invariant for the compiler: a scalar kept across the loop would be one more spilled SGPR) */
	const uint32_t sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sv);
	/* slice of instruction r = p ^ rot(8 r + sub) = p ^ ((sub >> 1) + 4 (r & 1)) = p ^ (sub >> 1) ^ 4 (r & 1): two values, for even and for odd r */
	const uint32_t sl0 = (uint32_t)(p ^ (sub >> 1)), sl1 = sl0 ^ 4u;
	const uint32_t oL[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
	if (first == 0) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
			const uint32_t slice = (r & 1) ? sl1 : sl0;
			__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(buckets + (size_t)(oL[r] + slice)), (void __attribute__((address_space(3))) *)(uintptr_t)(sbase + 1024u * r), 16, 0, BWB_GATHER_AUX);
		}
	}
	const uint32_t oU[4] = { b0.x, b0.y, b0.z, b0.w };
	const int nU_here = pi.nU - first; /* (wave-uniform: a scalar compare and branch per group of eight U rows) */
#pragma unroll
	for (int r = 0; r < NU / 8; r++) {
		const uint32_t slice = (r & 1) ? sl1 : sl0;
		if (nU_here > 8 * r) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(buckets + (size_t)(oU[r] + slice)), (void __attribute__((address_space(3))) *)(uintptr_t)(sbase + 8192u + 1024u * r), 16, 0, BWB_GATHER_AUX);
	}
	/* (Tried, session 12: the loads as structured-buffer loads - record = slice, idle owners out of range, no exec mask, three instructions
	 * a load instead of nine.  Correct on every test index and wrong at GRCh37 size: without swizzling the range check works on bytes,
	 * 32 bits of them, and the table has 13.7 GB.) */
	__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): the slices have landed in LDS - and so has every other load issued before them (the per-position
	                                       record, the heap entry a pop uncovered, the next list interval: all issued ahead of the gather) */
	asm volatile("" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}

/* Rank from a 64-character bucket (bwb_device.h): acc[j] = #j among the first n (0..32) characters of the sub-block whose planes are p,
 * j = 1..15: ONE masked pass. */
__device__ __forceinline__ void sub_pops16(const u32x4 p, int n, uint32_t acc[16]) {
	const uint32_t m = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
	const uint32_t a[4] = { ~p.x & ~p.y, p.x & ~p.y, ~p.x & p.y, p.x & p.y };
	const uint32_t m2 = m & ~p.z, m2p = m & p.z;
	const uint32_t b[4] = { m2 & ~p.w, m2p & ~p.w, m2 & p.w, m2p & p.w };
#pragma unroll
	for (int c = 1; c < 16; c++) acc[c] = (uint32_t)__popc(a[c & 3] & b[c >> 2]);
	acc[0] = 0;
}
/* what one side of a pair needs from its bucket row besides the counts: read BEFORE the row's upper half is overwritten */
struct SideBits {
	u32x4 planes;    /* of the sub-block the position is in */
	u32x4 mid;       /* the counts inside the first sub-block as bytes, zero when the position is in the first sub-block */
	uint32_t first;  /* first character of the enclosing 128-character block */
	int n;           /* characters of the sub-block to count: (off & 31) + 1 */
};
/* A row of the staging area as its readers name it: R = the row's byte address in LDS (128-byte aligned) + 16 rot(row); slice s is at
 * R ^ 16 s (see wave_gather), word w of the row's logical contents at R ^ 4 w. */
typedef uint32_t RowRef;
__device__ __forceinline__ RowRef row_ref(Lds<u32x4> stage, uint32_t row) { return (uint32_t)(uintptr_t)stage + (row << 7) + ((row << 3) & 0x70u); }
__device__ __forceinline__ Lds<u32x4> row_slice(RowRef R, uint32_t s16) { return (Lds<u32x4>)(uintptr_t)(R ^ s16); }
__device__ __forceinline__ void side_read(RowRef R, int off, SideBits &sb) {
	const uint32_t w16 = ((uint32_t)off >> 1) & 0x10u; /* 16 x (off >> 5), off < 64 */
	sb.planes = *row_slice(R, 0x40u | w16);
	const u32x4 md = *row_slice(R, 0x60u);
	const u32x4 z = { 0u, 0u, 0u, 0u };
	sb.mid = w16 ? md : z;
	sb.first = *(Lds<uint32_t>)(uintptr_t)(R ^ 0x70u);
	sb.n = (off & 31) + 1;
}
/* rel[j] = count of j before the bucket (slices 0-3 of `cnt_row`, count-slice order) + mid + pass, for j = 0..15 IN CODE ORDER -> slices
 * `dst0` .. `dst0 + 3` of the lane's own row (slice k holds the codes 4k .. 4k+3: kid_get finds code j with three instructions; round 3
 * left them in the count-slice order, whose address arithmetic cost more than the rank's adds); `quirk`: O_alphabet's view of 5, 9, 11, 13.
 * Every count slice is read before the first result is written: the row is source and destination. */
__device__ __forceinline__ void side_finish(RowRef cntR, const SideBits &sb, bool quirk, RowRef ownR, int dst0) {
	/* acc[c] = count slice + popcount in ONE v_bcnt_u32_b32 (it adds its second operand), then the mid byte with one SDWA add: two
	 * instructions per code after the mask where the compiler's own choice - v_bcnt(x, 0), a byte extract and a three-operand add - is
	 * three (round 5; the empty asm keeps the intermediate sum apart, so the adds cannot be regrouped) */
	const u32x4 p = sb.planes;
	const int n = sb.n;
	const uint32_t m = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
	const uint32_t a[4] = { ~p.x & ~p.y, p.x & ~p.y, ~p.x & p.y, p.x & p.y };
	const uint32_t m2 = m & ~p.z, m2p = m & p.z;
	const uint32_t b[4] = { m2 & ~p.w, m2p & ~p.w, m2 & p.w, m2p & p.w };
	const uint32_t md[4] = { sb.mid.x, sb.mid.y, sb.mid.z, sb.mid.w };
	uint32_t acc[16];
#pragma unroll
	for (int s = 0; s < 4; s++) { /* (slice s = the codes 2s, 2s+1, 2s+8, 2s+9) */
		const u32x4 q = *row_slice(cntR, (uint32_t)s << 4);
		const int c0 = 2 * s, c1 = 2 * s + 1, c2 = 2 * s + 8, c3 = 2 * s + 9;
		uint32_t t0 = (uint32_t)__popc(a[c0 & 3] & b[c0 >> 2]) + q.x, t1 = (uint32_t)__popc(a[c1 & 3] & b[c1 >> 2]) + q.y;
		uint32_t t2 = (uint32_t)__popc(a[c2 & 3] & b[c2 >> 2]) + q.z, t3 = (uint32_t)__popc(a[c3 & 3] & b[c3 >> 2]) + q.w;
		asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
		acc[c0] = t0 + (md[s] & 255u); acc[c1] = t1 + ((md[s] >> 8) & 255u);
		acc[c2] = t2 + ((md[s] >> 16) & 255u); acc[c3] = t3 + (md[s] >> 24);
	}
	{ /* O_alphabet's view of the codes 5, 9, 11, 13: 1 - [first char of the block == j] (bit j of ~(1 << first)); selects, no branch */
		const uint32_t nh = ~(1u << (sb.first & 31u));
		acc[5] = quirk ? ((nh >> 5) & 1u) : acc[5]; acc[9] = quirk ? ((nh >> 9) & 1u) : acc[9];
		acc[11] = quirk ? ((nh >> 11) & 1u) : acc[11]; acc[13] = quirk ? ((nh >> 13) & 1u) : acc[13];
	}
#pragma unroll
	for (int k = 0; k < 4; k++) *row_slice(ownR, (uint32_t)(dst0 + k) << 4) = u32x4{ acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3] };
}
/* index of code j in the count-slice order */
__device__ __forceinline__ constexpr int cslot(int j) { return 4 * ((j & 7) >> 1) + (j & 1) + 2 * (j >> 3); }

/* where a lane's children are after wave_children */
template <typename P> struct KidCtx {
	RowRef R;              /* the lane's row: relL in slices 0-3, relU in slices 4-7 (code order: slice k = codes 4k .. 4k+3) */
	Lds<P> baseL, baseU;   /* base-table rows of the two positions */
	int nvis;              /* rank-block visits of the pair by the SURVEY 8(d) rule: a position that is neither -1 nor length-1 counts one */
	bool qL, qU;           /* O_alphabet's view of the codes 5, 9, 11, 13 applies to this side: value = C[j] - [first char of the block == j] */
};
/* child j = [vL(j) + 1, vU(j)] */
template <typename P> __device__ __forceinline__ void kid_get(const KidCtx<P> &kc, Lds<P> s_base, int j, P &L, P &U) {
	/* code j: word j of the row's logical contents (slice j >> 2, component j & 3) is at R ^ 4 j; relU four slices on: bit 6 flipped */
	const uint32_t a = kc.R ^ ((uint32_t)j << 2);
	const uint32_t rl = *(Lds<uint32_t>)(uintptr_t)a, ru = *(Lds<uint32_t>)(uintptr_t)(a ^ 0x40u);
	/* (O_alphabet's view of the codes 5, 9, 11, 13 - value = C[j] - [first char of the block == j] - needs nothing here: side_finish left
	 * rel = 1 - [first == j] and the side's base row is one of the second table, whose entries for these codes are C[j] - 1: load_base2.
	 * Round 3 selected the row and the -1 per child: ten instructions; one extra row and a multiply-add: three, and 2 % slower than this) */
	(void)s_base;
	const P bL = kc.baseL[j], bU = kc.baseU[j];
	L = (P)(bL + (P)rl + 1); U = (P)(bU + (P)ru);
}

/* Children of the SA interval [iL, iU] of every lane of the wave: child j = [vL(j) + 1, vU(j)], j = 1..15.
 *   alpha == false: v(j) = C[j] + Occ(j, pos)                                      (O(), bwt.c:348-372)
 *   alpha == true : O_alphabet (bwt.c:374-438): codes 5, 9, 11, 13 are not counted, v(j) = C[j] - [first char of the
 *                   block == j] (bwt.c:427-435,780); exact in the special-cased positions -1 and length-1.
 * Which of the two a visit needs is known before the rank: alpha for an expansion (inexact_match.c:382-383), exact for
 * calculate_d / the exact tail.  Gathers (every lane of the wave takes part), ranks both sides from LDS, leaves the children
 * in relative form in the lane's row (kid_get).  Returns the bitmask of non-empty children (bits 1..15); n_bkt += buckets
 * fetched for the WAVE (a wave-uniform counter).
 * Side U of a pair that shares its bucket with side L (five in six) is relL + #j among the characters (L-1, U]: the same planes. */
template <typename P, int NU = NU_MAX>
__device__ __forceinline__ uint32_t wave_children(const uint4 *__restrict__ buckets, P last_row, bool need, P iL, P iU, bool alpha, Lds<P> s_base,
                                                  Lds<u32x4> stage, Lds<u32x4> zero_row, int lane, uint32_t &n_bkt, KidCtx<P> &kc) {
	PairInfo<P> pi;
	pair_setup<P>(last_row, need, (P)(iL - 1), iU, lane, pi);
	n_bkt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_bkt + (uint32_t)__popcll(wballot(pi.blkL != NONE32)) + (uint32_t)pi.nU)); /* (wave-uniform: the whole wave's buckets, added up by lane 0 at the end) */
	wave_gather<P, NU>(buckets, pi, 0, stage, lane);
	const RowRef ownR = row_ref(stage, (uint32_t)lane), zeroR = (uint32_t)(uintptr_t)zero_row;
	kc.R = ownR;
	kc.nvis = need ? (pi.regL ? 1 : 0) + (pi.regU ? 1 : 0) : 0;
	kc.qL = alpha && pi.regL; kc.qU = alpha && pi.regU;
	/* (qL / qU imply a superblock row, 0 .. BWB_NSB_MAX - 1: the second table has just those) */
	kc.baseL = s_base + pi.rowL * 16 + (kc.qL ? BWB_BASE_ROWS * 16 : 0); kc.baseU = s_base + pi.rowU * 16 + (kc.qU ? BWB_BASE_ROWS * 16 : 0);
	/* Both sides are independent one-pass ranks.  A lane's own row is source (counts 0-3, planes and mid counts 4-7: side L, and side U of
	 * a pair in one bucket) and destination (relL -> 0-3, relU -> 4-7): what side L needs from the upper half goes to registers first,
	 * then side U is finished (it still finds the counts in the lower half), then side L. */
	uint32_t ne = 0;
	{
		const bool haveL = pi.blkL != NONE32;
		const RowRef srcL = haveL ? ownR : zeroR;
		SideBits bl;
		side_read(srcL, pi.offL, bl);
		for (int first = 0;; first += NU) {
			if (first > 0) wave_gather<P, NU>(buckets, pi, first, stage, lane); /* (rare: more than NU lanes of the wave with a second bucket) */
			const bool fetched = pi.ku != NONE32;
			const bool now = fetched ? ((int)pi.ku >= first && (int)pi.ku < first + NU) : first == 0;
			if (now) {
				const uint32_t k = pi.ku - (uint32_t)first;
				const RowRef src = fetched ? row_ref(stage, 64u + k) : (pi.same ? ownR : zeroR);
				SideBits bu;
				side_read(src, pi.offU, bu);
				side_finish(src, bu, kc.qU, ownR, 4);
			}
			if (first == 0) {
				__builtin_amdgcn_sched_barrier(0);
				side_finish(srcL, bl, kc.qL, ownR, 0); /* (a lane whose U row comes in a later round keeps relL in its row meanwhile) */
				__builtin_amdgcn_sched_barrier(0);
			}
			if (first + NU >= pi.nU) break;
		}
		/* non-empty children, when both positions have the same base row: relU > relL (slice k holds the codes 4k .. 4k+3).  The mask is shifted
		 * in from the top, one code per compare + add-with-carry (ne = 2 ne + [relU > relL]), three compares in flight so that no consumer
		 * follows its producer by less than the two wait states a VALU-written scalar needs on gfx950: 32 instructions and no s_nop where the
		 * compiler's compare / select / or (and its nops) took about fifty */
#pragma unroll
		for (int k = 3; k >= 0; k--) {
			const u32x4 l = *row_slice(ownR, (uint32_t)k << 4), q = *row_slice(ownR, (uint32_t)(4 + k) << 4);
			unsigned long long sa_, sb_, st_;
			asm("v_cmp_gt_u32_e64 %[sa], %[qw], %[lw]\n\t"
			    "v_cmp_gt_u32_e64 %[sb], %[qz], %[lz]\n\t"
			    "v_cmp_gt_u32_e32 vcc, %[qy], %[ly]\n\t"
			    "v_addc_co_u32_e64 %[ne], %[st], %[ne], %[ne], %[sa]\n\t"
			    "v_cmp_gt_u32_e64 %[sa], %[qx], %[lx]\n\t"
			    "v_addc_co_u32_e64 %[ne], %[st], %[ne], %[ne], %[sb]\n\t"
			    "v_addc_co_u32_e32 %[ne], vcc, %[ne], %[ne], vcc\n\t"
			    "v_addc_co_u32_e64 %[ne], %[st], %[ne], %[ne], %[sa]"
			    : [ne] "+v"(ne), [sa] "=&s"(sa_), [sb] "=&s"(sb_), [st] "=&s"(st_)
			    : [qw] "v"(q.w), [lw] "v"(l.w), [qz] "v"(q.z), [lz] "v"(l.z), [qy] "v"(q.y), [ly] "v"(l.y), [qx] "v"(q.x), [lx] "v"(l.x)
			    : "vcc");
		}
	}
	const bool rows_differ = need && pi.rowL != pi.rowU;
	if (wany(rows_differ)) { /* a pair that straddles a superblock boundary or has a special position (the root's -1 / length-1): compare positions */
		if (rows_differ) {
			ne = 0;
#pragma unroll 1
			for (int j = 1; j < 16; j++) { P L, U; kid_get<P>(kc, s_base, j, L, U); ne |= (L <= U ? 1u : 0u) << j; }
		}
	}
	return need ? (ne & 0xFFFEu) : 0u;
}

/* ---- SA-interval list being built: add_sa_interval (align.c:93-110), tail in registers ------------ */
template <typename P> struct ListW {
	int T;        /* intervals so far, including the open tail */
	P tL, tU;
	P fL, fU;     /* the list's FIRST interval, once it is final (T >= 2): the step that reads this list next starts with it, and gets it from
	                 here instead of loading what this step has just stored (kl_search) */
};
/* the list being built is the one the current step does not read: buffer base + sel * cap (no pointer kept in registers).
 * One divergent region only - the store of a tail that cannot be merged -, everything else is selects: the loops that call this run
 * several trips per wave iteration with a few lanes each (kl_calc_d: 5.4, kl_search's exact steps: 2.6, tools/bbprof.py), and the nest
 * of four branches this was until round 4 cost more scalar mask bookkeeping per trip (25 instructions) than it did work. */
template <typename P> __device__ __forceinline__ void list_add(ListW<P> &l, Intv<P> *base, int sel, P L, P U, int cap) {
	const bool has = l.T != 0;
	const bool merge = has & (L == (P)(l.tU + 1));
	const bool flush = has & !merge;
	/* The tail goes to its slot WHATEVER happens to it (round 5): when it is merged, or when there is none yet (slot 0), the slot is written
	 * again before anything reads it - a list's readers take the intervals below the tail from memory and the tail from registers - and
	 * all but one append in two hundred flush anyway: the branch around the store was a divergent region per trip, 2.6 trips per iteration. */
	{ Intv<P> *buf = base + sel * cap; const int at = l.T > 0 ? l.T - 1 : 0; Intv<P> v; v.L = l.tL; v.U = l.tU; buf[at] = v; }
	const bool first = flush & (l.T == 1);
	l.fL = first ? l.tL : l.fL; l.fU = first ? l.tU : l.fU;
	l.tL = merge ? l.tL : L;
	l.tU = U;
	l.T += merge ? 0 : 1;
}
/* room for the children of one more interval (at most 15 appends)?  Checked once per interval, ahead of its appends: a list that comes
 * within 15 intervals of its capacity sends the read to the next scratch class a little early, which changes nothing but that. */
template <typename P> __device__ __forceinline__ bool list_full(const ListW<P> &l, int cap) { return l.T + 15 > cap; }

/* A read is finished: its status, hit count, offset and hit-log records are published BEFORE the slot's counter moves.  The
 * host polls that counter and copies the results on another stream while this kernel may still be running (a slot's parked reads
 * finish inside the NEXT slot's slice): plain stores can sit dirty in this XCD's L2 until the end of the kernel, so a release
 * fence (L2 write-back) comes first.  Once per read (tens of thousands of iterations): its cost does not show. */
__device__ __forceinline__ void publish_done(unsigned int *done) {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
	atomicAdd(done, 1u);
}

__device__ __forceinline__ uint32_t grab_read(const Work &wk) {
	const uint32_t w = atomicAdd(wk.counter, 1u); /* the compiler folds this into one atomic per wave */
	if (w >= wk.n_work) return NONE32;
	return wk.worklist ? wk.worklist[w] : w;
}

/* Loads that are ISSUED AHEAD of the iteration's gather and land under the gather's wait, IN PLACE, for the lanes named by `mask`.  The
 * destination register holds live values in the other lanes, so the compiler will not let an ordinary load inside a divergent branch write
 * it: it loads into a scratch register and copies under the exec mask - and the copy waits for the load on the spot, which puts the memory
 * round trip back in front of the gather.  Here the load is one statement in uniform control flow whose asm narrows the exec mask itself: to
 * the compiler a plain read-modify-write of `dst`, which stays where it is.  THE RULE: the compiler does not know that the register is in
 * flight - nothing may read it until a vmcnt(0) wait has been executed (kl_search: the gather's own wait, or the explicit one after it when
 * no lane of the wave needed a rank).  tools/check_prefetch_regs.py verifies on the ISA of every build that no instruction touches these
 * registers in between. */
__device__ __forceinline__ void prefetch128(u32x4 &dst, const void *p, unsigned long long mask) {
	unsigned long long sv;
	asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dwordx4 %[d], %[p], off\n\ts_mov_b64 exec, %[sv]" : [d] "+v"(dst), [sv] "=&s"(sv) : [p] "v"(p), [m] "s"(mask) : "memory", "scc");
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void prefetch64(u32x2 &dst, const void *p, unsigned long long mask) {
	unsigned long long sv;
	asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dwordx2 %[d], %[p], off\n\ts_mov_b64 exec, %[sv]" : [d] "+v"(dst), [sv] "=&s"(sv) : [p] "v"(p), [m] "s"(mask) : "memory", "scc");
}
__device__ __forceinline__ void prefetch32(uint32_t &dst, const void *p, unsigned long long mask) {
	unsigned long long sv;
	asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\tglobal_load_dword %[d], %[p], off\n\ts_mov_b64 exec, %[sv]" : [d] "+v"(dst), [sv] "=&s"(sv) : [p] "v"(p), [m] "s"(mask) : "memory", "scc");
}

/* an SA interval in the registers it is loaded into (16 bytes with 64-bit positions, 8 with 32-bit ones) */
template <typename P> struct IntvRegs;
template <> struct IntvRegs<uint64_t> {
	typedef u32x4 W;
	static __device__ __forceinline__ void fetch(W &d, const Intv<uint64_t> *p, unsigned long long m) { prefetch128(d, p, m); }
	static __device__ __forceinline__ uint64_t lo(const W w) { return ((uint64_t)w.y << 32) | w.x; }
	static __device__ __forceinline__ uint64_t hi(const W w) { return ((uint64_t)w.w << 32) | w.z; }
	static __device__ __forceinline__ W make(uint64_t L, uint64_t U) { return W{ (uint32_t)L, (uint32_t)(L >> 32), (uint32_t)U, (uint32_t)(U >> 32) }; }
};
template <> struct IntvRegs<uint32_t> {
	typedef u32x2 W;
	static __device__ __forceinline__ void fetch(W &d, const Intv<uint32_t> *p, unsigned long long m) { prefetch64(d, p, m); }
	static __device__ __forceinline__ uint32_t lo(const W w) { return w.x; }
	static __device__ __forceinline__ uint32_t hi(const W w) { return w.y; }
	static __device__ __forceinline__ W make(uint32_t L, uint32_t U) { return W{ L, U }; }
};
/* kl_calc_d's list being built: finished intervals are collected FOUR AT A TIME and stored as one aligned group (round 5: a 16-byte store
 * is a 32-byte write request to the memory fabric, 12 G of them per launch at GRCh37 scale; back-to-back stores into one 64-byte line merge) */
template <typename P> struct ListBuf { typename IntvRegs<P>::W b0, b1, b2, b3; };
template <typename P> __device__ __forceinline__ void list_store_group(const ListBuf<P> &wb, Intv<P> *buf, int t) { /* the group that holds interval t */
	typedef typename IntvRegs<P>::W W;
	W *g = (W *)(buf + (t & ~3));
	g[0] = wb.b0; g[1] = wb.b1; g[2] = wb.b2; g[3] = wb.b3;
}
template <typename P> __device__ __forceinline__ void list_add_buf(ListW<P> &l, ListBuf<P> &wb, Intv<P> *base, int sel, P L, P U, int cap) {
	typedef IntvRegs<P> IR;
	const bool has = l.T != 0;
	const bool merge = has && L == (P)(l.tU + 1);
	const bool flush = has && !merge;
	const int t = l.T - 1, q = t & 3;
	const typename IR::W v = IR::make(l.tL, l.tU);
	wb.b0 = (flush && q == 0) ? v : wb.b0; wb.b1 = (flush && q == 1) ? v : wb.b1;
	wb.b2 = (flush && q == 2) ? v : wb.b2; wb.b3 = (flush && q == 3) ? v : wb.b3;
	if (flush && q == 3) list_store_group<P>(wb, base + sel * cap, t);
	const bool first = flush && l.T == 1;
	l.fL = first ? l.tL : l.fL; l.fU = first ? l.tU : l.fU;
	l.tL = merge ? l.tL : L;
	l.tU = U;
	l.T += merge ? 0 : 1;
}
/* the position is finished: the intervals of the last, incomplete group go to memory (indices 0 .. T-2 are the finished ones) */
template <typename P> __device__ __forceinline__ void list_flush_buf(const ListW<P> &l, const ListBuf<P> &wb, Intv<P> *base, int sel, int cap) {
	if (l.T >= 2 && ((l.T - 2) & 3) != 3) list_store_group<P>(wb, base + sel * cap, l.T - 2);
}

/* ============================================================================================
 * The calculate_d table (round 6).  The first K steps of calculate_d (inexact_match.c:171-254) depend on nothing but the index and the LAST K
 * bases of the sequence: the interval list, the restart count z, the summed width of the last step and the K bytes {min(z,127) | width-equal}
 * after K steps are a function of that K-mer.  At GRCh37 scale those steps are where the lists are long - every code string with one or two
 * IUPAC codes that is compatible with the bases occurs somewhere in 6.85 G rows until the suffix is about 16 long - and hold most of
 * kl_calc_d's rank visits.  The table holds that state for all 4^K K-mers (K = 12: 16.7 M entries of 32 bytes + their interval lists), built
 * once per context, level by level: the state of a k-mer is ONE step from the state of its (k-1)-mer (k_dtab_level), so the whole table
 * costs about as much as calculate_d of the 4^K last steps.  kl_calc_d starts a read (and its seed) from the table; what it then computes,
 * stores and hands to kl_search is bit for bit what the K steps would have produced.  288 GB of HBM are what makes this a table.
 *
 * Entry (two uint4): a = { list offset low 32 bits, offset bits 32..39 | z << 8 | valid << 15 | T << 16, nm (summed width of step K-1, the
 * 32-bit wrap of the reference's int), rank-block visits of the K steps by the SURVEY 8(d) rule }, b = the K bytes (12 at most).
 * An entry is invalid when a list on the way outgrew the build's buffers: such a read takes the ordinary path. */
#define DTAB_KMAX 12
template <typename P> struct DTab {
	const uint4 *ent;       /* [4^K][2]; NULL: no table */
	const Intv<P> *pool;
	int K;
	uint32_t nm1[4];        /* summed width of the FIRST step by base (the level-1 entries' nm): the width-equal bit of a lookup that follows a restart */
};
__device__ __forceinline__ uint32_t dtab_T(const uint4 a) { return a.y >> 16; }
__device__ __forceinline__ bool dtab_valid(const uint4 a) { return (a.y >> 15) & 1u; }
__device__ __forceinline__ uint32_t dtab_z(const uint4 a) { return (a.y >> 8) & 127u; }
__device__ __forceinline__ unsigned long long dtab_off(const uint4 a) { return (unsigned long long)a.x | ((unsigned long long)(a.y & 255u) << 32); }

/* one level of the table: entry `id` of level k (k-mer = its (k-1)-mer `id mod 4^(k-1)` followed by base `id >> 2(k-1)`) from the entries of
 * level k-1.  One k-mer per lane; the parent's list is read in place, the children go to the lane's list scratch (kl_calc_d's, both halves)
 * and from there to the level's pool at an offset taken from `bump`. */
template <typename P>
__global__ __launch_bounds__(LANE_BLOCK, CALCD_WAVES_PER_SIMD) void k_dtab_level(DevIndex ix, int multiref, int level, const uint4 *pent, const Intv<P> *ppool, uint4 *oent, Intv<P> *opool,
                                                                                  unsigned long long *bump, unsigned long long pool_cap, LaneScratch sc) {
	extern __shared__ __align__(16) unsigned char smem_[];
	const LdsBytes smem = (LdsBytes)(uintptr_t)(((uint32_t)(uintptr_t)(LdsBytes)smem_ + 127u) & ~127u);
	P *s_base = (P *)smem;
	const int lane = (int)(threadIdx.x & 63u);
	LdsBytes wlds = smem + CALCD_WAVES_OFF + (threadIdx.x >> 6) * WAVE_LDS_BYTES_NU(CALCD_NU);
	Lds<u32x4> stage = (Lds<u32x4>)wlds, zero_row = (Lds<u32x4>)(smem + CALCD_ZERO_OFF);
	const Lds<P> sb = (Lds<P>)smem;
	if (threadIdx.x < 32) ((Lds<uint32_t>)zero_row)[threadIdx.x] = 0u;
	load_base<P>(s_base, ix);
	const uint32_t slot = blockIdx.x * LANE_BLOCK + threadIdx.x;
	Intv<P> *lbuf = (Intv<P> *)sc.lists_d + (size_t)slot * 2 * sc.lcap;
	const int cap = (int)(2 * sc.lcap);
	const uint4 *__restrict__ buckets = ix.buckets;
	const P last_row = (P)(ix.length - 1);
	const unsigned long long n_tasks = 1ull << (2 * level), nl = (unsigned long long)gridDim.x * LANE_BLOCK;
	const unsigned long long pmask = (1ull << (2 * (level - 1))) - 1ull;
	for (unsigned long long base = 0; base < n_tasks; base += nl) { /* (whole waves iterate together: the gather is a wave's work) */
		const unsigned long long id = base + slot;
		const bool have = id < n_tasks;
		const unsigned long long pid = have ? (id & pmask) : 0ull;
		const int c = (int)(id >> (2 * (level - 1))) & 3;
		const uint4 pa = pent[2 * pid], pb = pent[2 * pid + 1];
		const bool pvalid = have && dtab_valid(pa);
		const Intv<P> *plist = ppool + dtab_off(pa);
		int T = pvalid ? (int)dtab_T(pa) : 0, s = 0;
		ListW<P> nx; nx.T = 0; nx.tL = nx.tU = 0; nx.fL = nx.fU = 0;
		uint32_t nm = 0, r_vis = 0, nbk = 0;
		bool ovf = false;
		while (wany(s < T)) {
			const bool need = s < T;
			Intv<P> iv; iv.L = 0; iv.U = 0;
			if (need) iv = plist[s];
			KidCtx<P> kc;
			uint32_t ne = wave_children<P, CALCD_NU>(buckets, last_row, need, iv.L, iv.U, false, sb, stage, zero_row, lane, nbk, kc);
			if (need) {
				r_vis += (uint32_t)kc.nvis;
				ne &= multiref ? member_mask(c) : single_mask_codes(c);
				if (list_full<P>(nx, cap)) { ovf = true; T = 0; ne = 0; }
				while (ne) { /* ascending code order == nucl_bases_table order (io.h:102-106) */
					const int j = __ffs((int)ne) - 1;
					ne &= ne - 1;
					P L, U;
					kid_get<P>(kc, sb, j, L, U);
					nm += (uint32_t)(U - L + 1);
					list_add<P>(nx, lbuf, 0, L, U, cap);
				}
				s++;
			}
		}
		if (have) {
			bool valid = pvalid && !ovf;
			uint32_t z = dtab_z(pa);
			int newT = nx.T;
			P tL = nx.tL, tU = nx.tU;
			if (newT == 0) { newT = 1; tL = 0; tU = last_row; z++; nm = (uint32_t)ix.length; } /* no matches: restart with the full interval (:240-244) */
			const int k = level - 1; /* the D index of this step */
			const uint32_t byte = (z > 127u ? 127u : z) | ((k > 0 && nm == pa.z) ? 0x80u : 0u);
			unsigned long long off = 0;
			if (valid) {
				off = atomicAdd(bump, (unsigned long long)newT);
				if (off + (unsigned long long)newT > pool_cap) valid = false;
			}
			if (valid) {
				Intv<P> *dst = opool + off;
				for (int t = 0; t + 1 < newT; t++) dst[t] = lbuf[t];
				Intv<P> tl; tl.L = tL; tl.U = tU;
				dst[newT - 1] = tl;
			}
			uint32_t db[3] = { pb.x, pb.y, pb.z };
			db[k >> 2] = (db[k >> 2] & ~(255u << (8 * (k & 3)))) | (byte << (8 * (k & 3)));
			oent[2 * id] = make_uint4((uint32_t)off, (uint32_t)(off >> 32) | (z << 8) | ((valid ? 1u : 0u) << 15) | ((uint32_t)newT << 16), nm, pa.w + r_vis);
			oent[2 * id + 1] = make_uint4(db[0], db[1], db[2], 0u);
		}
	}
}

/* ============================================================================================
 * k_calc_d (one read per lane)
 * ========================================================================================== */
template <typename P>
__global__ __launch_bounds__(LANE_BLOCK, CALCD_WAVES_PER_SIMD) void kl_calc_d(DevIndex ix, Batch b, Work wk, KParams kp, LaneScratch sc, int32_t *dbgD, int32_t *dbgDs,
                                                        uint32_t dbg_ld, uint32_t dbg_lds, unsigned long long *stats, DTab<P> dt) {
	extern __shared__ __align__(16) unsigned char smem_[];
	const LdsBytes smem = (LdsBytes)(uintptr_t)(((uint32_t)(uintptr_t)(LdsBytes)smem_ + 127u) & ~127u); /* (rows of the staging area are 128-byte aligned: RowRef; the host asks for 128 bytes more) */
	P *s_base = (P *)smem;
	const int lane = (int)(threadIdx.x & 63u);
	LdsBytes wlds = smem + CALCD_WAVES_OFF + (threadIdx.x >> 6) * WAVE_LDS_BYTES_NU(CALCD_NU);
	Lds<u32x4> stage = (Lds<u32x4>)wlds, zero_row = (Lds<u32x4>)(smem + CALCD_ZERO_OFF);
	const Lds<P> sb = (Lds<P>)smem; /* the base table again, as an LDS pointer */
	if (threadIdx.x < 32) ((Lds<uint32_t>)zero_row)[threadIdx.x] = 0u;
	load_base<P>(s_base, ix);
	const uint32_t slot = blockIdx.x * LANE_BLOCK + threadIdx.x;
	Intv<P> *lbase = (Intv<P> *)sc.lists_d + (size_t)slot * 2 * sc.lcap;
	const int cap = (int)sc.lcap;
	const uint4 *__restrict__ buckets = ix.buckets;
	const P last_row = (P)(ix.length - 1);

	bool active = false, done = false;
	uint32_t rid = 0;
	int len = 0, phase = 0, plen = 0, r = 0, z = 0, s = 0, curT = 0, cursel = 0;
	Intv<P> nxi; nxi.L = nxi.U = 0; bool nxi_valid = false; /* the FIRST interval of the list a position has just completed (registers: ListW::fL / fU) */
	/* The intervals of the current list that are in memory are read FOUR AT A TIME (round 5): one aligned 64-byte group (32 bytes with 32-bit
	 * positions) per four iterations instead of a 16-byte load per iteration.  A scattered 16-byte load is a whole 64-byte request to the
	 * memory fabric (tools/pmc_traffic.sh calibration), the L1 is swept by the gather between two iterations, and kl_calc_d runs at 84 % of
	 * the bandwidth a streaming read reaches with its traffic 1.44 x its bucket bytes - 0.7 list requests per iteration were most of that
	 * excess (profiles/r4_c3_pmc.json).  gI = the cached group, cg = its number (interval index >> 2), -1: none. */
	typedef IntvRegs<P> IR;
	typename IR::W g0 = {}, g1 = {}, g2 = {}, g3 = {};
	int cg = -1;
	int c = 4, cnext = 4; /* seq[r] and seq[r - 1]: loaded once per position, one position ahead (round 2 loaded seq[r] in every iteration) */
	P cL = 0, cU = 0; /* tail (last interval) of the current list */
	ListW<P> nx; nx.T = 0; nx.tL = nx.tU = 0; nx.fL = nx.fU = 0;
	ListBuf<P> wb; wb.b0 = wb.b1 = wb.b2 = wb.b3 = typename IntvRegs<P>::W{};
	int32_t nm = 0, prev_nm = 0;
	uint32_t bacc = 0, cntN = 0; /* (bacc: the bases of the record being filled) */
	unsigned long long vis = 0;
	uint32_t r_vis = 0, n_bkt = 0;
	const uint8_t *seq = b.reads;
	const Intv<P> *curb = lbase; /* the current list: one half of the lane's buffer - or, for the step that follows a table lookup, the table's own list */
	/* calculate_d is in the table's state - ONE interval, the whole index - at the start of a phase (the read, then its seed) and after every
	 * restart (:240-244: no match, or an N in the read - for two reads in three at GRCh37 scale, every substitution is one).  When the next K
	 * bases are four-letter ones and at least one position follows them, the K steps are looked up: the K bytes go to the records (z counted
	 * on from the present z; for the read itself also the bases), z, the summed width of step K-1 and the visit count are taken over, and the
	 * step after them reads the entry's list in place.  `r` = the position about to be processed.  Not with the debug arrays (they want every
	 * step's width: bwb_hip_calc_d). */
	auto from_table = [&]() -> bool {
		if (!dt.ent || dbgD || r < dt.K) return false;
		unsigned long long idx = 0;
		bool ok = true;
		for (int t = 0; t < dt.K; t++) { const uint32_t ch = seq[r - t]; ok &= ch <= 3u; idx |= (unsigned long long)(ch & 3u) << (2 * t); }
		if (!ok) return false;
		const uint4 ea = dt.ent[2 * idx], eb = dt.ent[2 * idx + 1];
		const int T = (int)dtab_T(ea);
		if (!dtab_valid(ea) || T + 15 > cap) return false;
		uint8_t *rec = b.dbuf + (size_t)rid * b.dstride;
		const uint32_t dbw[3] = { eb.x, eb.y, eb.z };
		for (int t = 0; t < dt.K; t++) {
			const int k = plen - 1 - (r - t); /* the D index of position r - t */
			const uint32_t tb = (dbw[t >> 2] >> (8 * (t & 3))) & 255u;
			const uint32_t zz = (uint32_t)z + (tb & 127u);
			uint32_t byte = (zz > 127u ? 127u : zz) | (tb & 128u);
			/* (the table's first step has no width-equal bit - D[0] has none, :246 -; after a restart the step compares with the restart's width) */
			if (t == 0 && k > 0 && dt.nm1[seq[r] & 3u] == (uint32_t)prev_nm) byte |= 128u;
			rec_put(rec, rec_count((uint32_t)len), phase ? 8 : 0, phase ? k + (len - kp.seed_length) : k, byte);
			if (!phase) { /* (as at the end of a computed position, below) */
				const int i1 = k + 1;
				bacc |= (uint32_t)seq[len - i1] << (4 * (i1 & 3));
				if ((i1 & 3) == 3 || i1 == len) { *(uint16_t *)(rec + REC_BYTES * (i1 >> 2) + 6) = (uint16_t)bacc; bacc = 0; }
			}
		}
		z += (int)dtab_z(ea); prev_nm = (int32_t)ea.z; nm = 0; r_vis += ea.w;
		curb = dt.pool + dtab_off(ea); curT = T; s = 0; cg = -1; cursel = 0; nx.T = 0;
		{ const Intv<P> tl = curb[T - 1]; cL = tl.L; cU = tl.U; }
		if (T >= 2) nxi = curb[0];
		nxi_valid = T >= 2;
		r -= dt.K; c = seq[r]; cnext = r >= 1 ? seq[r - 1] : 4;
		return true;
	};

	for (;;) {
		if (!active && !done) {
			rid = grab_read(wk);
			if (rid == NONE32) done = true;
			else {
				len = b.lens[rid];
				const bool unrep = len == BAD_LEN; /* a read the kernels cannot represent (host: slot_upload): empty record */
				if (unrep) len = 0;
				r_vis = 0;
				seq = b.reads + (size_t)rid * b.stride;
				phase = 0; plen = len; r = len - 1; z = 0; s = 0; cursel = 0; nm = 0; prev_nm = 0; bacc = 0; cntN = 0;
				{ /* the read's records (bwb_kernels.h: rec_put) start as zeros with their tags: D[-2], D[-1], the seed's DS below its first position */
					uint4 *rz = (uint4 *)(b.dbuf + (size_t)rid * b.dstride);
					for (uint32_t m = 0; m < rec_count((uint32_t)len); m++) rz[m] = make_uint4(0u, 0u, 0u, m << 16);
				}
				c = len > 0 ? seq[len - 1] : 4; cnext = len > 1 ? seq[len - 2] : 4;
				cL = 0; cU = last_row; curT = 1; nxi_valid = false; cg = -1;
				nx.T = 0; curb = lbase;
				active = len > 0;
				if (!(kp.seed_length && len > kp.seed_length)) {
					/* D_seed is only computed when len > seed_length (inexact_match.c:141-143); otherwise the reference reads whatever
					 * its buffer holds: the bounds of the last longer read before this one in the file (serial path, one buffer:
					 * inexact_match.c:35) - k_dseed_inherit copies those in afterwards - or the calloc'd zeros written here (num_diff 0,
					 * equal widths) when there is no such read. */
					uint8_t *rec = b.dbuf + (size_t)rid * b.dstride;
					for (int k = -2; k < len; k++) /* (down to k = -2: the hit check, i = 0, of a read shorter than the seed consults D_seed too, :324-328) */
						if (k - (len - kp.seed_length) >= 0) rec_put(rec, rec_count((uint32_t)len), 8, k, 0x80u);
				}
				if (active && kp.use_precalc) /* -P: a read with an N in the last 12 bases of rc is dropped before calculate_d (inexact_match.c:129-136) */
					for (int k = 0; k < PRECALC_LEN; k++) if (seq[k] > 3) active = false;
				if (!active) { b.dbuf[(size_t)rid * b.dstride + b.dstride - 4] = 0; b.status[rid] = ST_OK; }
				else (void)from_table();
			}
		}
		if (wall(done)) break;
		P iL = 0, iU = 0;
		/* (the cached group is only ever touched here and by the prefetch below, in straight-line code at the loop's top level: with its uses
		 * inside the nest of branches the register allocator kept a second copy of it and moved one into the other while the prefetch was in
		 * flight - tools/check_prefetch_regs.py, which the Makefile runs on every build) */
		const int gq = s & 3;
		const typename IR::W gw = gq == 0 ? g0 : (gq == 1 ? g1 : (gq == 2 ? g2 : g3));
		if (active) {
			if (c > 3 && phase == 0) cntN++;
			if (c <= 3) {
				if (s == curT - 1) { iL = cL; iU = cU; }
				else if (nxi_valid) { iL = nxi.L; iU = nxi.U; } /* the new list's first interval (registers) */
				else if ((s >> 2) == cg) { iL = IR::lo(gw); iU = IR::hi(gw); } /* from the group fetched ahead of an earlier iteration's gather */
				else { const Intv<P> v = curb[s]; iL = v.L; iU = v.U; asm volatile("" :: "v"(iL), "v"(iU)); } /* (never in the steady state: waited for inside the branch) */
			}
		}
		{ /* the group of the interval of the position's NEXT iteration, when that is one of the list in memory and not in the cached group:
		   * issued now, IN PLACE (prefetch128: nothing looks at g0..g3 before the gather's wait below), and landing under this iteration's
		   * gather.  The cached group is not needed any more then: the next interval is the first of another group. */
			const bool want = active && c <= 3 && s + 1 < curT - 1 && ((s + 1) >> 2) != cg;
			const unsigned long long mw = wballot(want);
			/* (not under `if (mw)`: a branch around the in-place loads makes the compiler merge a loaded and a not-loaded version of the four
			 * registers behind it - copies of registers that are in flight; with an empty mask the loads are no-ops) */
			const Intv<P> *g = curb + ((s + 1) & ~3);
			IR::fetch(g0, g, mw); IR::fetch(g1, g + 1, mw); IR::fetch(g2, g + 2, mw); IR::fetch(g3, g + 3, mw);
			cg = want ? (s + 1) >> 2 : cg;
		}
		const bool need = active && c <= 3;
		uint32_t nbk = 0;
		KidCtx<P> kc;
		uint32_t ne = wave_children<P, CALCD_NU>(buckets, last_row, need, iL, iU, false, sb, stage, zero_row, lane, nbk, kc); /* every lane of the wave loads */
		n_bkt += nbk;
		if (!active) { nxi_valid = false; continue; }
		bool ovf = false;
		if (c <= 3) {
			r_vis += (uint32_t)kc.nvis;
			ne &= kp.multiref ? member_mask(c) : single_mask_codes(c); /* -S: the base's own code only (inexact_match.c:176-206) */
			if (list_full<P>(nx, cap)) { ovf = true; ne = 0; }
			while (ne) { /* children in ascending code order == nucl_bases_table order (io.h:102-106) */
				const int j = __ffs((int)ne) - 1;
				ne &= ne - 1;
				P L, U;
				kid_get<P>(kc, sb, j, L, U);
				nm += (int32_t)(uint32_t)(U - L + 1);
				list_add_buf<P>(nx, wb, lbase, cursel ^ 1, L, U, cap);
			}
			s++;
		}
		if (ovf) { b.status[rid] = ST_D_OVF; active = false; nxi_valid = false; continue; }
		if (c > 3 || s >= curT) {
			/* position finished: swap lists (inexact_match.c:234-237) */
			if (c <= 3) list_flush_buf<P>(nx, wb, lbase, cursel ^ 1, cap);
			cursel ^= 1; curb = lbase + cursel * cap;
			curT = (c > 3) ? 0 : nx.T; cL = nx.tL; cU = nx.tU;
			if (curT >= 2) { nxi.L = nx.fL; nxi.U = nx.fU; } /* (the new list's first interval: from registers - list_add -, not from what this step has just stored) */
			nx.T = 0; s = 0; cg = -1;
			const bool restarted = curT == 0;
			if (restarted) { /* no matches: restart with the full interval (inexact_match.c:240-244) */
				cL = 0; cU = last_row; curT = 1; z++;
				nm = (int32_t)(uint32_t)ix.length;
			}
			const int k = plen - 1 - r; /* D index */
			{ /* D[k] is what an entry with e->i == k+1 reads as D[i-1] and one with e->i == k+2 as D[i-2] (:317,399-405) */
				const uint32_t byte = (uint32_t)((z > 127 ? 127 : z) | ((k > 0 && nm == prev_nm) ? 0x80 : 0));
				uint8_t *rec = b.dbuf + (size_t)rid * b.dstride;
				rec_put(rec, rec_count((uint32_t)len), phase ? 8 : 0, phase ? k + (len - kp.seed_length) : k, byte); /* (seed: its index k at the read's position k + len - seed_length) */
				if (!phase) { /* seq[len - i1], i1 = k + 1: the complement of the base an entry at i1 extends with (io.c:502-504); four positions share 16 bits */
					const int i1 = k + 1;
					bacc |= (uint32_t)c << (4 * (i1 & 3));
					if ((i1 & 3) == 3 || i1 == len) { *(uint16_t *)(rec + REC_BYTES * (i1 >> 2) + 6) = (uint16_t)bacc; bacc = 0; }
				}
			}
			if (dbgD) {
				int32_t *dst = phase ? dbgDs + ((size_t)rid * dbg_lds + k) * 2 : dbgD + ((size_t)rid * dbg_ld + k) * 2;
				dst[0] = z; dst[1] = nm;
			}
			prev_nm = nm; nm = 0; r--;
			if (r >= 0) { c = cnext; cnext = r >= 1 ? seq[r - 1] : 4; }
			if (r < 0) {
				if (dbgD) { /* D[readLen] (inexact_match.c:249-250) */
					int32_t *dst = phase ? dbgDs + ((size_t)rid * dbg_lds + plen) * 2 : dbgD + ((size_t)rid * dbg_ld + plen) * 2;
					dst[0] = z + 1; dst[1] = 0;
				}
				if (phase == 0 && kp.seed_length && len > kp.seed_length) { /* inexact_match.c:141-143 */
					phase = 1; plen = kp.seed_length; r = plen - 1; z = 0; prev_nm = 0;
					cL = 0; cU = last_row; curT = 1;
					c = seq[r]; cnext = r >= 1 ? seq[r - 1] : 4;
					(void)from_table();
				} else {
					b.dbuf[(size_t)rid * b.dstride + b.dstride - 4] = (uint8_t)(cntN > 255 ? 255 : cntN);
					*(uint32_t *)(b.dbuf + (size_t)rid * b.dstride + b.dstride - 8) = r_vis; /* work done for this read: a cheap predictor of search cost */
					b.status[rid] = ST_OK;
					vis += r_vis;
					active = false;
				}
			} else if (restarted) (void)from_table(); /* (the state is the table's again: the next K positions by lookup) */
		}
		/* the interval of the next iteration: the list's tail and a new list's first interval are in registers, the others come from the group cache */
		nxi_valid = active && c <= 3 && s == 0 && curT >= 2;
	}
	if (vis) { atomicAdd(&stats[STAT_VIS_SINGLE], vis); atomicAdd(&stats[STAT_VIS_CALCD], vis); }
	if (lane == 0 && n_bkt) atomicAdd(&stats[STAT_BKT_CALCD], (unsigned long long)n_bkt);
}

/* ============================================================================================
 * k_search (one read per lane)
 * ========================================================================================== */
/* Test build only (`make testlib`): heap entries store 64-bit positions plus this constant, so that the three high bits of L
 * and U that a 16-byte entry packs into its last word (positions >= 2^32: only a > 4.3 G-row index has them) are exercised
 * by a small index.  0 in the product. */
#ifndef BWB_TEST_POS_BIAS
#define BWB_TEST_POS_BIAS 0ull
#endif
template <typename P> __device__ __forceinline__ P pos_enc(P v) { return sizeof(P) == 8 ? (P)((uint64_t)v + (uint64_t)BWB_TEST_POS_BIAS) : v; }
template <typename P> __device__ __forceinline__ P pos_dec(P v) { return sizeof(P) == 8 ? (P)((uint64_t)v - (uint64_t)BWB_TEST_POS_BIAS) : v; }

/* heap entry.  NARROW (max_gapo <= 1): 16 bytes {L lo, U lo, i|mm|go|ge, state|alen<<2|run<<10|L hi<<26|U hi<<29};
 * WIDE: 32 bytes {L lo, U lo, i|mm|go|ge, state|alen<<2|L hi<<26|U hi<<29} {eight gap runs} (round 5: -o up to 8; rounds 3-4 kept the
 * positions as two 64-bit words and had room for four runs).
 * runs: one 16-bit word per gap open: start | len<<8 | isD<<15, 0xFFFF = unused. */
template <typename P> struct LEntry {
	P L, U;
	uint32_t f;        /* i | mm<<8 | go<<16 | ge<<24 */
	uint32_t sa;       /* state | alen<<2 */
	uint32_t runsLo, runsHi;   /* gap runs 0..3 */
	uint32_t runs2Lo, runs2Hi; /* gap runs 4..7 (32-byte entries only) */
};

/* Score-bucketed LIFO heap (inexact_match.h:17-34, inexact_match.c:510-610): every bucket is a chain of 64-slot
 * chunks.  Chunk header (slot 0): .x = next chunk of the lane's free list, .y = state word of the bucket before this
 * chunk was started (so a chunk may be left partially filled), .z = next chunk in the lane's private chain / excess chain / the block's recycle stack.
 * A bucket state word is chunk<<6 | fill (fill = index of the top entry, 1..63), NONE32 when empty. */
template <typename P, bool WIDE> struct LHeap {
	uint4 *pool;
	unsigned int *pool_bump;
	uint32_t pool_cap;
	uint32_t *bstate;      /* this lane's row: bstate[s] */
	uint32_t nslots;
	uint32_t fhead;        /* chunks emptied by pops during this read (LIFO through header .x): the list's head.  With FHEAD_TAKEN set: the chunk
	                          an allocation took from the list - the new head is in ITS header and is fetched, in place, with the next
	                          iteration's prefetches (LHeap::prefetch), ahead of the gather; until then the list counts as empty */
#define FHEAD_TAKEN 0x80000000u /* (a chunk number has 26 bits; NONE32 has this bit too: test for NONE32 first) */
	uint32_t pblk, pused, keep;  /* the lane's private run of `keep` consecutive chunks starts at chunk (pblk + lane in wave) * keep (pblk: wave-uniform,
	                                the wave's first lane in its region); how many of them this read has taken */
	uint32_t pshared;            /* first chunk of the region's shared part (after every lane's private run) */
	uint32_t xhead;              /* chunks the current read took beyond the private chain: the chain's head (the chunk taken last).  Its tail
	                                (the first one: a constant once set) and the chain's length live in memory - words 0 and 1 of the lane's
	                                `xs` area, which the callers name with a callable evaluated only on these rare paths: registers */
	Lds<unsigned long long> blockfree; /* head of this block's stack of recycled chunks (LDS): version<<32 | chunk */
	Lds<unsigned int> nfree;     /* chunks on that stack */
	uint64_t neW;          /* non-empty buckets, as a window above the cached one: bit k = bucket cb + k.  Entries are popped in
	                          non-decreasing score order and a child's score exceeds its parent's by at most one penalty, so every
	                          non-empty bucket lies in [cb, cb + 63] - with penalties up to 63; above that see `far` below */
	int cb;                /* bucket whose state is cached in registers = score of the entry last popped */
	uint32_t cst;
	/* The buckets an expansion of an entry of bucket cb pushes to besides cb itself are cb + mm_score (mismatches), cb + gapo_score
	 * (gap opened from STATE_M) and cb + gape_score (gap extended): their states live in registers too while cb is the cached bucket -
	 * they only ever grow meanwhile (nothing but cb is popped).  Round 3 loaded a side bucket's state from memory after the rank,
	 * whenever the expansion pushed to it, and stored it back: a dependent memory round trip in almost every iteration of a wave
	 * (some lane pushes a gap or a mismatch) and 1.6 requests per lane iteration.  Memory is brought up to date when the cached bucket
	 * changes (switch_cache), which happens a few dozen times per read.  Penalties that coincide share a register (side_of). */
	uint32_t stX, stGo, stGe;
	int pX, pGo, pGe;      /* the three penalties (wave-uniform copies of the kernel parameters) */
	int nbk;               /* number of buckets: a side bucket beyond the score range does not exist */
	int pXv, pGov, pGev, nbkv; /* the same four numbers once more, in VECTOR registers, for per-lane arithmetic (the scalar copies decide the wave-uniform branches): see kl_search */
	int num_entries;
	/* Register mirror of the TWO entries on top of bucket cb's stack, in the packed form they have in memory (tw: the top - 16 bytes, 32
	 * when WIDE; sw: the one below it, 16-byte entries only).  Why two, and why packed: a pop takes the top from `tw`; what it uncovers
	 * comes from `sw` when that is valid, else it is loaded from memory AT ONCE - the load is issued before the iteration's gather and
	 * lands under the gather's wait; the words are not touched until the next pop unpacks them, so nothing waits for them in between
	 * (round 3 reloaded the top at the end of the iteration and used it at the start of the next: an exposed memory round trip in every
	 * iteration of a wave, because some lane always needed it).  When a match child is then pushed - it becomes the top and is popped
	 * next - the uncovered entry moves down to `sw` instead of being thrown away, and serves the pop after next.  The match child itself
	 * exists in `tw` only (its slot is reserved, never written); `sw` always has a copy in memory. */
	u32x4 tw, tw1, sw;
	bool top_valid, sec_valid;
	uint32_t cprev;        /* header word .y of the chunk that holds bucket cb's top = the bucket's state before that chunk was started: what
	                          cst becomes when the chunk's last entry is popped (0 = not known, fetch it then).  Kept in a register so that the
	                          pop that crosses a chunk boundary (one in 63 per lane, some lane in every other iteration of a wave) does not
	                          wait for memory; the next chunk's word is fetched at that moment, 63 pops ahead of its use. */

	/* last word of a 16-byte entry: state|aln_length (10 bits), the one gap run (16), bits 32..34 of L and of U (the
	 * superblock table covers 2^34 BWT characters, bwb_device.h) */
	static __device__ __forceinline__ uint32_t pack_w(P L, P U, uint32_t sa, uint32_t runsLo) {
		uint32_t w = sa | ((runsLo & 0xFFFFu) << 10);
		if (sizeof(P) == 8) w |= ((uint32_t)((uint64_t)L >> 32) << 26) | ((uint32_t)((uint64_t)U >> 32) << 29);
		return w;
	}
	__device__ __forceinline__ uint4 *chunk_ptr(uint32_t c) const { return pool + (size_t)c * (CHUNK_SLOTS * (WIDE ? 2 : 1)); }
	__device__ __forceinline__ void reset() { /* (fhead is NONE32 already: set when the previous read finished) */ pused = 0; neW = 0; cb = 0; cst = NONE32; num_entries = 0; top_valid = false; sec_valid = false; cprev = 0; stX = stGo = stGe = NONE32; }
	/* which register holds the state of the side bucket at distance `pen` (a wave-uniform penalty) from cb: 0 = cb itself (zero penalty),
	 * 1 = stX, 2 = stGo, 3 = stGe.  Equal penalties share the first register of the order X, Go, Ge. */
	__device__ __forceinline__ int side_of(int pen) const { return pen == 0 ? 0 : (pen == pX ? 1 : (pen == pGo ? 2 : 3)); }
	/* the side registers -> memory (when the cached bucket changes, when the read is parked) */
	__device__ __forceinline__ void side_flush() const {
		if (pX != 0 && cb + pXv < nbkv) bstate[cb + pXv] = stX;
		if (side_of(pGo) == 2 && cb + pGov < nbkv) bstate[cb + pGov] = stGo;
		if (side_of(pGe) == 3 && cb + pGev < nbkv) bstate[cb + pGev] = stGe;
	}
	__device__ __forceinline__ void side_load() {
		stX = (pX != 0 && cb + pXv < nbkv) ? bstate[cb + pXv] : NONE32;
		stGo = (side_of(pGo) == 2 && cb + pGov < nbkv) ? bstate[cb + pGov] : NONE32;
		stGe = (side_of(pGe) == 3 && cb + pGev < nbkv) ? bstate[cb + pGev] : NONE32;
	}
	/* Penalties above 63 (WIDE instantiation only: the host routes such parameters to the 32-byte-entry kernels, whatever -o is): a child's
	 * bucket can lie beyond the window.  Such a bucket is simply not marked - its state is in a side register or in memory - and the window
	 * is topped up from the bucket states in memory whenever it moves (switch_cache, after the side registers have been written back); when
	 * the window is empty the states beyond it are scanned (best).  `far` is wave-uniform; the 16-byte-entry kernels do not have the code. */
	bool far;
	__device__ __forceinline__ void mark(int s) { if (!WIDE || !far || s - cb < 64) neW |= 1ull << (s - cb); }
	__device__ __forceinline__ void unmark(int s) { if (!WIDE || !far || s - cb < 64) neW &= ~(1ull << (s - cb)); }
	__device__ __forceinline__ int best(int nb) {
		if (neW) return cb + __ffsll((long long)neW) - 1;
		if (WIDE && far) { /* (rare: nothing within 63 of the cached bucket) */
			side_flush();
			for (int k = cb + 64; k < nbk; k++) if (bstate[k] != NONE32) return k;
		}
		return nb;
	}
	__device__ __forceinline__ void switch_cache(int s) {
		if (s == cb) return;
		bstate[cb] = cst;
		side_flush();
		/* (the new cached bucket is very often the old mismatch bucket: its state is at hand) */
		const int d = s - cb;
		const bool have = d == pXv || d == pGov || d == pGev;
		const uint32_t vX = stX, vGo = stGo, vGe = stGe; /* (values first: a conditional between two members is a conditional between two addresses, and keeps the whole struct in memory) */
		const uint32_t fwd = d == pXv ? vX : (d == pGov ? vGo : vGe);
		neW = (WIDE && far && d >= 64) ? 0ull : neW >> d; /* (s > cb: the cached bucket is empty and nothing lies below it) */
		cb = s; cst = have ? fwd : bstate[s];
		if (WIDE && far) { /* buckets that have come into the window's range: their states are in memory (side_flush above) */
			for (int j = d >= 64 ? 0 : 64 - d; j < 64 && s + j < nbk; j++) if (j == 0 || bstate[s + j] != NONE32) neW |= 1ull << j;
		}
		side_load();
		/* (the states just loaded are waited for HERE, inside the branch: left pending, they make the compiler put an `s_waitcnt vmcnt(0)`
		 * where the branches meet again - in front of every pop of the wave, which then sits out the previous iteration's stores) */
		asm volatile("" :: "v"(cst), "v"(stX), "v"(stGo), "v"(stGe));
		top_valid = false; sec_valid = false; cprev = 0;
	}
	/* Chunk sources, in order: chunks this read has already emptied (fhead); the lane's private run of `keep` consecutive
	 * chunks at the start of its region (a counter: an ordinary read allocates without touching memory); the block's stack
	 * of chunks recycled by finished reads (lock-free: versioned head, so a pop cannot be fooled by a chunk that left and
	 * came back); the shared part of the region (atomic bump).  What a read takes beyond its private run is threaded on its
	 * excess chain and goes back to the block stack in one push when the read ends, so the pool holds what the reads in
	 * flight need, not the worst case every lane has ever seen. */
	template <bool FRESH = false, typename XS> __device__ __forceinline__ uint32_t alloc(bool &ovf, XS xs) {
		/* The list of emptied chunks: its head is in a register, the rest is linked through the chunks' headers.  Taking the head
		 * leaves the list EMPTY for the rest of the iteration (fhead = the taken chunk | FHEAD_TAKEN); the new head (a load from the
		 * taken chunk's header) is fetched with the next iteration's prefetches, in place, and lands under that iteration's gather wait:
		 * round 3 loaded it here, inside this divergent branch, and the wave sat out that round trip in almost every iteration (some
		 * lane allocates).  FRESH: the first allocation of a read (loop top): the list is empty by construction. */
		if (!FRESH && fhead < FHEAD_TAKEN) { const uint32_t c = fhead; fhead = c | FHEAD_TAKEN; return c; }
		if (pused < keep) { /* (laying the runs of a block out chunk-major, for page locality, measured no different) */
			uint32_t z = 0u;
			asm volatile("" : "+v"(z)); /* (the lane number is made here, from an opaque zero: as a loop invariant it would take a register across the loop - or a scratch slot) */
			const uint32_t lid = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
			return (pblk + lid) * keep + pused++;
		}
		uint32_t c = NONE32;
		unsigned long long old = __hip_atomic_load(blockfree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		while ((uint32_t)old != NONE32) {
			const uint32_t top = (uint32_t)old;
			const uint32_t nxt = __hip_atomic_load(&chunk_ptr(top)[0].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned long long want = (((old >> 32) + 1) << 32) | nxt;
			if (__hip_atomic_compare_exchange_strong(blockfree, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { c = top; __hip_atomic_fetch_add(nfree, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
		}
		if (c == NONE32) {
			c = pshared + atomicAdd(pool_bump, 1u);
			if (c >= pool_cap) { ovf = true; return 0; }
		}
		__hip_atomic_store(&chunk_ptr(c)[0].z, xhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		uint32_t *x = xs();
		if (xhead == NONE32) x[0] = c; /* the chain's tail */
		atomicAdd(x + 1, 1u);          /* its length (no return value: nothing waits for it) */
		xhead = c;
		return c;
	}
	/* end of a read: hand the chunks it took beyond the private chain to the block */
	template <typename XS> __device__ __forceinline__ uint32_t release_excess(XS xs) { /* returns how many chunks the chain held */
		if (xhead == NONE32) return 0u;
		uint32_t *x = xs();
		const uint32_t xtail = x[0], xcnt = x[1];
		x[1] = 0u;
		unsigned long long old = __hip_atomic_load(blockfree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		for (;;) {
			__hip_atomic_store(&chunk_ptr(xtail)[0].z, (uint32_t)old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); /* the link is in L2 before the head can name this chain */
			const unsigned long long want = (((old >> 32) + 1) << 32) | xhead;
			if (__hip_atomic_compare_exchange_strong(blockfree, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
		}
		__hip_atomic_fetch_add(nfree, xcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		xhead = NONE32;
		return xcnt;
	}
	/* makes room for k (<= 63) more entries on a bucket whose state is st; returns the state to push from */
	template <bool FRESH = false, typename XS> __device__ __forceinline__ uint32_t reserve(uint32_t st, int k, bool &ovf, XS xs) {
		if (k > 0 && (st == NONE32 || (int)(st & 63u) + k > CHUNK_SLOTS - 1)) {
			const uint32_t c = alloc<FRESH>(ovf, xs);
			if (ovf) return st;
			chunk_ptr(c)[0].y = st;
			return c << 6;
		}
		return st;
	}
	/* an entry in the form it has in memory: w0 (and w1 when WIDE) */
	static __device__ __forceinline__ void pack(P L, P U, uint32_t f, uint32_t sa, uint32_t runsLo, uint32_t runsHi, u32x4 &w0, u32x4 &w1, uint32_t runs2Lo = ~0u, uint32_t runs2Hi = ~0u) {
		L = pos_enc<P>(L); U = pos_enc<P>(U);
		if (WIDE) {
			w0 = u32x4{ (uint32_t)L, (uint32_t)U, f, pack_w(L, U, sa & 0x3FFu, 0u) };
			w1 = u32x4{ runsLo, runsHi, runs2Lo, runs2Hi };
		} else w0 = u32x4{ (uint32_t)L, (uint32_t)U, f, pack_w(L, U, sa, runsLo) };
	}
	static __device__ __forceinline__ void unpack(const u32x4 w0, const u32x4 w1, LEntry<P> &e) {
		e.L = (P)w0.x; e.U = (P)w0.y; e.f = w0.z;
		if (sizeof(P) == 8) { e.L |= (P)((uint64_t)((w0.w >> 26) & 7u) << 32); e.U |= (P)((uint64_t)(w0.w >> 29) << 32); }
		if (WIDE) {
			e.sa = w0.w & 0x3FFu;
			e.runsLo = w1.x; e.runsHi = w1.y; e.runs2Lo = w1.z; e.runs2Hi = w1.w;
		} else {
			e.sa = w0.w & 0x3FFFFFFu; /* (state | aln_length << 2 | the one gap run << 10: ERUNS_LO; one register, not two) */
			e.runsLo = 0xFFFFFFFFu; e.runsHi = 0xFFFFFFFFu; e.runs2Lo = 0xFFFFFFFFu; e.runs2Hi = 0xFFFFFFFFu; /* (not used with 16-byte entries: the run is in e.sa) */
		}
		e.L = pos_dec<P>(e.L); e.U = pos_dec<P>(e.U);
	}
	__device__ __forceinline__ void store_packed(uint32_t st, const u32x4 w0, const u32x4 w1) const {
		typedef __attribute__((address_space(1))) u32x4 *G4;
		G4 p = (G4)(uintptr_t)(pool + (size_t)st * (WIDE ? 2 : 1)); /* (a state word is the slot's index in the pool: chunk << 6 | fill) */
		p[0] = w0;
		if (WIDE) p[1] = w1;
	}
	/* the entry at state st -> tw (tw1): issued, not waited for - the words are first looked at by the pop that unpacks them */
	__device__ __forceinline__ void load_top(uint32_t st) {
		typedef const __attribute__((address_space(1))) u32x4 *G4;
		G4 p = (G4)(uintptr_t)(pool + (size_t)st * (WIDE ? 2 : 1));
		tw = p[0];
		if (WIDE) tw1 = p[1];
		top_valid = true;
	}
	/* the entry (L, U, f, sa, runs) becomes the top of the cached bucket, in registers only (the caller has reserved its slot and counted it) */
	__device__ __forceinline__ void set_top(P L, P U, uint32_t f, uint32_t sa, uint32_t runsLo, uint32_t runsHi) { pack(L, U, f, sa, runsLo, runsHi, tw, tw1); top_valid = true; } /* (entries without a gap: runs 4..7 unused) */
	/* Pops the top entry of the cached bucket cb (the best non-empty one).  n_ld += 1 when an entry is fetched from memory.  What the pop
	 * uncovers comes from the second mirror register, else from memory: pf_top = its state word, pf_hdr = the chunk whose header word is
	 * wanted for cprev (NONE32: nothing to fetch) - the caller issues both with prefetch() where the lanes of the wave have met again. */
	__device__ __forceinline__ void pop(LEntry<P> &e, uint32_t &n_ld, uint32_t &pf_top, uint32_t &pf_hdr, uint32_t &pf_free) {
		if (!top_valid) { /* (only after the cached bucket changed, or when something else than a match was pushed on top of it) */
			load_top(cst); n_ld++;
			/* this rare load is waited for HERE, inside its branch: a wait after the branches have met again would be executed in every
			 * iteration, and - vmcnt counts in issue order - would have to be a wait for everything */
			asm volatile("" :: "v"(tw.x), "v"(tw.y), "v"(tw.z), "v"(tw.w));
			if (WIDE) asm volatile("" :: "v"(tw1.x), "v"(tw1.y), "v"(tw1.z), "v"(tw1.w));
		}
		unpack(tw, tw1, e);
		asm volatile("" : "+v"(e.L), "+v"(e.U), "+v"(e.f), "+v"(e.sa)); /* (the entry is in its own registers before the load below can be issued into tw) */
		if ((cst & 63u) == 1u) { /* the last entry of its chunk: the chunk goes to the lane's free list, the bucket continues in the chunk before */
			typedef __attribute__((address_space(1))) uint32_t *G1;
			G1 hd = (G1)(uintptr_t)chunk_ptr(cst >> 6);
			uint32_t pv = cprev;
			if (pv == 0u) { pv = hd[1]; asm volatile("" :: "v"(pv)); } /* (not known: only the first crossing after the cached bucket changed; waited for inside the branch) */
			pf_free = cst >> 6; /* (goes onto the lane's list of emptied chunks after the gather's wait - give_back() -: the list's head may be in flight) */
			cst = pv;
			if (pv == NONE32) unmark(cb);
			cprev = 0u;
			if (pv != NONE32) pf_hdr = pv >> 6; /* (wanted 63 pops from now) */
		} else cst--;
		if (!WIDE && sec_valid) { tw = sw; top_valid = true; sec_valid = false; }
		else if (cst != NONE32) { pf_top = cst; top_valid = true; n_ld++; }
		else top_valid = false;
		num_entries--;
	}
	/* the chunk a pop emptied goes onto the lane's list (after the gather's wait: a head that was being fetched is there) */
	__device__ __forceinline__ void give_back(uint32_t c) {
		if (c != NONE32) { ((__attribute__((address_space(1))) uint32_t *)(uintptr_t)chunk_ptr(c))[0] = fhead; fhead = c; }
	}
	/* issues what pop() asked for (every lane of the wave calls this, in uniform control flow) */
	__device__ __forceinline__ void prefetch(uint32_t pf_top, uint32_t pf_hdr) {
		const unsigned long long mt = wballot(pf_top != NONE32), mh = wballot(pf_hdr != NONE32);
		{ /* (not under `if (mt)`: some lane of the wave uncovers an entry in nearly every iteration, and a branch around an in-place load makes
		   * the compiler merge a loaded and a not-loaded version of the registers behind it; with an empty mask the load is a no-op) */
			{
				const uint4 *src = pool + (size_t)pf_top * (WIDE ? 2 : 1); /* (a state word is the slot's index in the pool) */
				prefetch128(tw, src, mt);
				if (WIDE) prefetch128(tw1, src + 1, mt);
			}
		}
		if (mh) prefetch32(cprev, (const uint32_t *)chunk_ptr(pf_hdr) + 1, mh);
		/* the new head of the list of emptied chunks, when an allocation of the previous iteration took the old one */
		const bool taken = fhead != NONE32 && (fhead & FHEAD_TAKEN) != 0u;
		const unsigned long long mf = wballot(taken);
		if (mf) prefetch32(fhead, (const uint32_t *)chunk_ptr(fhead & ~FHEAD_TAKEN), mf);
	}
};

/* sum over the wave of a small per-lane count (< 32), as a wave-uniform value: five ballots (the lanes must have converged) */
__device__ __forceinline__ uint32_t wave_sum5(uint32_t v) {
	uint32_t t = 0;
#pragma unroll
	for (int b = 0; b < 5; b++) t += (uint32_t)__popcll(wballot(((v >> b) & 1u) != 0u)) << b;
	return t;
}

#define LMODE_POP 0
#define LMODE_EXACT 1
#define STATE_M 0 /* align.h:16-18 */
#define STATE_I 1
#define STATE_D 2
/* Heap-only fourth state: a DELETION GROUP.  An expansion that may open or extend a deletion pushes one deletion child per
 * non-empty code (:448-463) - up to 15 entries that differ in their SA interval only and sit, in sequence, on the gap bucket.
 * At GRCh37 scale they are more than half of all pushes, and at least five in six are never popped (the search ends first: a gap
 * costs more than three mismatches).  So the children are not stored: ONE entry with this state, the parent's interval and the
 * children's fields stands in their place, counted as the children it represents (num_entries, pushes).  When it reaches the top
 * of its bucket, the lane ranks the parent's interval again and replaces the group by its children - same bucket, same
 * position, code order - without counting a pop or a visit; from then on they are ordinary STATE_D entries.  LIFO order inside
 * the bucket, the entry count that max_entries is checked against, and every pop the reference makes are unchanged. */
#define STATE_GROUP 3
#ifdef BWB_STAMPS
#define STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

/* Diagnostic build (`make hist`, tools_exp/libbwbble_hip_hist.so): wave-aggregated event counters of the search loop - what the
 * iterations are (expansion / exact step / pruned / hit), how wide the expanded intervals are, how many children they have, what
 * is pushed where.  HIST(k, cond) counts the lanes for which cond holds; HISTW(k, v) adds a wave-uniform value.  Off in the product. */
#ifdef BWB_HIST
#define HIST(k, cond) do { hist[k] += (unsigned long long)__popcll(wballot(cond)); } while (0)
#define HISTW(k, v) do { hist[k] += (unsigned long long)(v); } while (0)
#else
#define HIST(k, cond) do { } while (0)
#define HISTW(k, v) do { } while (0)
#endif
enum { H_ITER = 0, H_POP, H_POP_FROM_MIRROR, H_POP_GAPPED, H_PRUNED, H_HIT, H_EXACT_START, H_EXPAND, H_EXACT_STEP, H_NEED_RANK, H_SAME_BKT, H_TWO_BKT,
       H_W1, H_W2, H_W4, H_W8, H_W32, H_W128, H_WBIG, H_NE0, H_NE1, H_NE2, H_NE3_4, H_NE5_8, H_NE9, H_PUSH_GAP, H_PUSH_MIS, H_PUSH_MATCH,
       H_DEL_OK, H_MM_OK, H_INS_OK, H_TOP_RELOAD, H_WAVE_ITERS, H_WAVE_GAPLOOP, H_WAVE_MISLOOP, H_WAVE_MATCHLOOP, H_WAVE_ANY_TWO_BKT, H_WAVE_ANY_WIDE8,
       H_WAVE_NREQ_LE16, H_ALPHA, H_EXACT_MULTI, H_FINISH, H_ALLOC, H_WAVE_ANY_EXACT, H_WAVE_ANY_EXPAND, H_WAVE_ALL_EXACT, H_EXP_SAME_W1, H_EXP_SAME_W2_4, H_PH_GAP, H_PH_MIS, H_PH_POP, H_EXPAND_AFTER_HIT, H_N };

/* Slices.  One launch of kl_search = one slice of the stream of batches.  Per-read work is heavy-tailed, so a
 * launch that runs until its last read is done ends with ever fewer busy lanes.  Instead, when the cursor of the batch runs
 * out (wk.suspend) a wave whose lane could not get a new read PARKS the reads its other lanes are working on - every
 * register of the per-lane state machine goes to the lane's save area, the heap, lists and hits are in the lane's global
 * scratch anyway - and leaves; the next launch (which starts the next batch) resumes them in the same lanes.  A parked read
 * belongs to an earlier slot than the reads its wave starts next, hence the per-lane `myslot` and the slot table. */
/* MULTI: the multi-genome alphabet (is_multiref; false = -S).  A template parameter, not a look at MULTI: the choice sits in every
 * trip of the child loops (the -S children are rows 1..4 mapped back to their codes), and the loop is bound by instruction issue. */
template <typename P, bool WIDE, bool MULTI>
__global__ __launch_bounds__(LANE_BLOCK, LANE_WAVES_PER_SIMD) void kl_search(DevIndex ix, const SlotDesc *__restrict__ descs, Work wk, KParams kp, LaneScratch sc, unsigned long long *stats) {
	/* The kernel arguments that only rare paths use (starting, parking and finishing a read, the statistics) are read from the kernarg
	 * segment where they are used - the pointer goes through an empty asm, so the loads cannot be hoisted out of the loop - instead of
	 * living in scalar registers across it (NOT the hit list's: the end of an exact tail is rare for a lane but happens in most iterations
	 * of a wave, and a scalar load there is an exposed round trip to the scalar cache - measured 3.7 % slower at GRCh37 scale,
	 * profiles/r4_ab_steps.txt): the loop keeps about a hundred wave-uniform values and lane masks alive, and what
	 * does not fit 102 scalar registers is moved to and from VGPR lanes (v_writelane / v_readlane) around every use.  KArgsT mirrors the
	 * kernel's parameter list (same order, by value: the kernarg segment's layout). */
	struct KArgsT { DevIndex ix; const SlotDesc *descs; Work wk; KParams kp; LaneScratch sc; unsigned long long *stats; };
#define KARGS ({ const __attribute__((address_space(4))) KArgsT *p_ = (const __attribute__((address_space(4))) KArgsT *)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p_)); p_; })
#define R_descs (KARGS->descs)
#define R_wk(f) (KARGS->wk.f)
#define R_stats (KARGS->stats)
#define R_sc(f) (KARGS->sc.f)
	extern __shared__ __align__(16) unsigned char smem_[];
	const LdsBytes smem = (LdsBytes)(uintptr_t)(((uint32_t)(uintptr_t)(LdsBytes)smem_ + 127u) & ~127u); /* (rows of the staging area are 128-byte aligned: RowRef; the host asks for 128 bytes more) */
	P *s_base = (P *)smem;
	const int lane = (int)(threadIdx.x & 63u);
	const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* (wave-uniform: a scalar) */
	LdsBytes wlds = smem + LDS_WAVES_OFF + wave_in_block * WAVE_LDS_BYTES;
	Lds<u32x4> stage = (Lds<u32x4>)wlds, zero_row = (Lds<u32x4>)(smem + LDS_ZERO_OFF);
	const Lds<P> sb = (Lds<P>)smem; /* the base table again, as an LDS pointer */
	__shared__ unsigned long long s_blockfree;
	__shared__ unsigned int s_active, s_nfree, s_left; /* reads in flight in this block; chunks on its recycle stack; waves that have left */
	__shared__ unsigned int s_need_sum, s_need_cnt;    /* chunks (in units of 16) the reads finished by this block took, and how many reads: admission */
	if (threadIdx.x == 0) {
		const uint32_t *bs = R_sc(blocksave) + (size_t)blockIdx.x * 4;
		const bool keep = wk.resume != 0;
		s_blockfree = keep ? ~(((unsigned long long)bs[1] << 32) | bs[0]) : ~0ull;
		s_nfree = keep ? bs[2] : 0u;
		s_active = 0; s_left = 0; s_need_sum = 0; s_need_cnt = 0;
	}
	/* where every slot's records are (a lane's read may belong to an earlier slot than the one this launch feeds from): in LDS, so that the
	 * lane needs no 64-bit pointer to its read's records across the loop - its slot and read numbers name them (registers) */
	__shared__ unsigned long long s_cnt[LANE_BLOCK / 64]; /* per wave: heap entries stored | fetched << 32 */
	if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = 0ull;
	__shared__ unsigned long long s_dbuf[BWB_MAX_SLOTS];
	__shared__ unsigned int s_dstride[BWB_MAX_SLOTS];
	if (threadIdx.x < BWB_MAX_SLOTS) { s_dbuf[threadIdx.x] = (unsigned long long)(uintptr_t)descs[threadIdx.x].b.dbuf; s_dstride[threadIdx.x] = descs[threadIdx.x].b.dstride; }
	if (threadIdx.x < 32) ((Lds<uint32_t>)zero_row)[threadIdx.x] = 0u;
	load_base2<P>(s_base, ix);
	/* The lane's hit list and save area are addressed from its slot number where they are used (rare paths); its bucket states and its
	 * lists - every iteration - through pointers that live across the loop (see below). */
	const uint32_t slot = blockIdx.x * LANE_BLOCK + threadIdx.x;
	uint32_t slotv = slot;
	/* The bases and sizes of the lane's scratch areas that the loop uses in (almost) every iteration, made OPAQUE scalar values: with more
	 * wave-uniform values alive than scalar registers, the compiler drops kernel arguments and loads them again from the kernarg segment
	 * where they are used - an s_load and a full wait on the scalar cache, five times per iteration in the first round-4 kernel
	 * (SQ_INSTS_SMEM, profiles/r4_c3_pmc_sq.txt).  A value it cannot see through is kept, or parked in a VGPR lane (one v_readlane). */
	typedef __attribute__((address_space(1))) unsigned char *GlobalBytes; /* (global pointers: a pointer made from an integer would be a flat one) */
	GlobalBytes sc_lists = (GlobalBytes)sc.lists, sc_alns = (GlobalBytes)sc.alns, sc_bstate = (GlobalBytes)sc.bstate;
	uint32_t sc_lcap = sc.lcap, sc_acap = sc.acap, sc_brow = sc.brow;
	/* (round 5) ... and held in VECTOR registers: the loop has more wave-uniform values than scalar registers, the compiler parks the loop
	 * invariants among them in lanes of a VGPR and fetches them with a v_readlane (+ hazard nops) in front of every use - 300 of them in the
	 * loop's code; the kernel needs 152 of the 168 vector registers that three waves per SIMD allow, so the invariants that only per-lane
	 * arithmetic reads live in the sixteen spare ones, where a vector instruction reads them for nothing.  The addresses of the lane's
	 * bucket states and lists are kept as pointers for the same reason (rounds 3-4 rebuilt them from the slot number where they were used,
	 * when registers were short): loop 5 613 -> 5 251 instructions, 299 -> 96 v_readlane, 87 -> 36 s_nop. */
	asm volatile("" : "+v"(sc_lists), "+v"(sc_alns), "+v"(sc_bstate), "+v"(sc_lcap), "+v"(sc_acap), "+v"(sc_brow));
	Intv<P> *const lbase = (Intv<P> *)(unsigned char *)sc_lists + (size_t)slotv * 2 * sc_lcap;
#define myalns ((uint4 *)(unsigned char *)sc_alns + (size_t)slotv * sc_acap * ALN_U4)
#define mysave (R_sc(save) + (size_t)slotv * SAVE_U4)
	auto xs = [&]() -> uint32_t * { return (uint32_t *)(mysave + 15); }; /* the tail and the length of the lane's excess chain (LHeap::alloc) */
	const int lcap = (int)sc_lcap;
	/* The alignment parameters as OPAQUE scalars: seen through, the compiler drops them when it runs out of scalar registers and loads them
	 * again from the kernarg segment inside the loop - 8.5 s_load + full lgkmcnt waits per wave iteration in the first round-5 builds
	 * (tools/bbprof.py), each a round trip to the scalar cache in the path of a whole wave.  A value it cannot see through is kept, or parked
	 * in a VGPR lane and fetched with one v_readlane. */
	KParams kq = kp;
#ifndef BWB_NO_KQ
	asm volatile("" : "+s"(kq.max_diff), "+s"(kq.max_gapo), "+s"(kq.max_gape), "+s"(kq.max_entries), "+s"(kq.mm_score), "+s"(kq.gapo_score), "+s"(kq.gape_score));
	asm volatile("" : "+s"(kq.seed_length), "+s"(kq.max_diff_seed), "+s"(kq.max_best), "+s"(kq.no_indel_length), "+s"(kq.num_buckets), "+s"(kq.use_precalc));
#endif
	const int nb = kq.num_buckets;
	const uint4 *__restrict__ buckets = ix.buckets;
	const P last_row = (P)(ix.length - 1);

	LHeap<P, WIDE> h;
	const uint32_t region = blockIdx.x % sc.n_regions;
	h.pblk = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x / sc.n_regions) * LANE_BLOCK + (threadIdx.x & ~63u)));
	h.pshared = ((gridDim.x + sc.n_regions - 1) / sc.n_regions) * LANE_BLOCK * sc.keep;
	h.pool = sc.pool + (size_t)region * sc.region_u4; h.pool_bump = sc.pool_bump + region * 16; h.pool_cap = sc.pool_cap; h.bstate = (uint32_t *)(unsigned char *)sc_bstate + (size_t)slotv * sc_brow; h.nslots = sc.nslots;
	h.xhead = NONE32; h.nfree = (Lds<unsigned int>)&s_nfree; h.keep = sc.keep; h.blockfree = (Lds<unsigned long long>)&s_blockfree;
	h.pX = kq.mm_score; h.pGo = kq.gapo_score; h.pGe = kq.gape_score; h.nbk = nb;
	h.pXv = h.pX; h.pGov = h.pGo; h.pGev = h.pGe; h.nbkv = nb;
	if (!WIDE) asm volatile("" : "+v"(h.pXv), "+v"(h.pGov), "+v"(h.pGev), "+v"(h.nbkv)); /* (the 32-byte-entry kernels have no registers to spare: there these stay scalars) */
	h.far = WIDE && (kq.mm_score > 63 || kq.gapo_score > 63 || kq.gape_score > 63);
#ifdef BWB_NO_PHANTOM /* (A/B: the round-5 behaviour - every pushed entry is stored) */
	const bool ph_ok = false;
#else
	const bool ph_ok = nb <= 256 && !h.far; /* count-only pushes (see the expansion); wave-uniform */
#endif
	h.fhead = NONE32;
	h.reset();

	bool active = false, done = false;
	uint32_t rid = 0;
	/* four small per-read numbers in one register: the read's length (8 bits), the difference budget (max_diff after the first hit, :336;
	 * 8 bits), the slot the read belongs to (4 bits), best_score (from bit 20: up to 1 024 buckets) */
	uint32_t rdw = wk.slot << 16;
#define rd_len ((int)(rdw & 255u))
#define rd_max_diff ((int)((rdw >> 8) & 255u))
#define rd_myslot ((rdw >> 16) & 15u)
#define rd_best_score ((int)(rdw >> 20))
#define SET_LEN(v) (rdw = (rdw & ~255u) | (uint32_t)(v))
#define SET_MAX_DIFF(v) (rdw = (rdw & ~0xFF00u) | ((uint32_t)(v) << 8))
#define SET_MYSLOT(v) (rdw = (rdw & ~0xF0000u) | ((uint32_t)(v) << 16))
#define SET_BEST_SCORE(v) (rdw = (rdw & 0xFFFFFu) | ((uint32_t)(v) << 20))
	bool exact_mode = false;                                /* LMODE_EXACT: the lane is in an exact tail (or seeding with -P) */
	int num_best = 0, n_alns = 0;
	int r = 0, s = 0, curT = 0;                             /* exact-tail state; it ends after rc[r_stop], r_stop = seeding ? len - 12 : 0 */
	bool cursel = false;                                    /* which of the two lists the step reads; it builds the other */
	Intv<P> nxi; nxi.L = nxi.U = 0; bool nxi_valid = false; /* exact tail: the next interval of a multi-interval list, fetched ahead */
	bool seeding = false;                                   /* -P: the exact steps under way build the read's first heap entries */
	uint32_t nxw = 0;                                       /* exact tail: summed width of the intervals added to the next list so far (wrapping, like the
	                                                           reference's int num_best sum :350-352) */
	ListW<P> nx; nx.T = 0; nx.tL = nx.tU = 0; nx.fL = nx.fU = 0;
	LEntry<P> e; e.L = e.U = 0; e.f = 0; e.sa = 0; e.runsLo = e.runsHi = e.runs2Lo = e.runs2Hi = ~0u;
	/* the popped entry's gap runs: four 16-bit words with 32-byte entries; with 16-byte entries the single run travels in bits 10..25 of e.sa */
#define ERUNS_LO (WIDE ? e.runsLo : (0xFFFF0000u | ((e.sa >> 10) & 0xFFFFu)))
#define ERUNS_HI (WIDE ? e.runsHi : 0xFFFFFFFFu)
#define ERUNS2_LO (WIDE ? e.runs2Lo : 0xFFFFFFFFu)
#define ERUNS2_HI (WIDE ? e.runs2Hi : 0xFFFFFFFFu)
	/* the tail (last interval) of the current list of an exact tail lives in the registers of the popped entry's interval: the tail starts as
	 * that interval (:345-347), and the entry's interval is not looked at again once its exact tail has begun (registers decide whether three
	 * waves fit a SIMD) */
#define cL e.L
#define cU e.U
	h.tw = h.tw1 = h.sw = u32x4{ 0u, 0u, 0u, 0u };
#define e_score (h.cb) /* the score of the entry being worked on = the bucket it was popped from: the cached one, which does not move until the next pop */
	u32x4 rec = { 0u, 0u, 0u, 0xFFFF0000u }; /* the record loaded last; its tag (high half of .w) says which four positions it serves: a match child is at
	                                             i - 1 and popped next, the intervals of an exact step share a position - most iterations find their record here */
	uint32_t r_vis_s = 0, r_vis_a = 0, r_pop = 0, r_push = 0; /* per read; committed (one atomic each, straight to the statistics) only when the read completes */
	uint32_t n_iter = 0, w_iter = 0;                          /* iterations of this lane / of this wave in this launch */
	uint32_t n_bkt = 0, n_rec = 0; /* wave-uniform: buckets fetched, records loaded (heap entries stored / fetched: s_cnt) */
	uint32_t acc_st = 0, acc_ld = 0; /* per lane, for the whole launch: heap entries stored / fetched (summed over the wave once, at the end) */
	bool parked = false;
#ifdef BWB_STAMPS
	unsigned long long seg[16] = { 0 }, tlast = __builtin_amdgcn_s_memtime();
#endif
#ifdef BWB_HIST
	unsigned long long hist[H_N] = { 0 };
	unsigned long long hl_gap = 0, hl_mis = 0, hl_match = 0, hl_phg = 0, hl_phx = 0; /* per-lane sums */
#endif
	if (wk.resume && (mysave[0].x & 1u)) {
		/* ---- resume the read this lane parked at the end of the previous slice ---- */
		const uint4 a0 = mysave[0], a1 = mysave[1], a2 = mysave[2], a3 = mysave[3], a4 = mysave[4], a5 = mysave[5], a6 = mysave[6],
		            a7 = mysave[7], a8 = mysave[8], a9 = mysave[9], a10 = mysave[10], a11 = mysave[11], a12 = mysave[12], a13 = mysave[13];
		auto p64 = [](uint32_t lo, uint32_t hi) { return (P)(((uint64_t)hi << 32) | lo); };
		auto v4 = [](const uint4 a) { return u32x4{ a.x, a.y, a.z, a.w }; };
		const uint32_t fl = a0.x;
		exact_mode = ((fl >> 1) & 1u) != 0; cursel = ((fl >> 2) & 1u) != 0; seeding = (fl >> 3) & 1u; nxi_valid = (fl >> 4) & 1u; h.top_valid = (fl >> 5) & 1u;
		h.sec_valid = (fl >> 6) & 1u;
		rid = a0.y; rdw = a0.z; num_best = (int)a0.w;
		n_alns = (int)a1.x; r = (int)a1.y; s = (int)a1.z; curT = (int)a1.w;
		nx.T = (int)a2.x; nxw = a2.y; h.cprev = a2.z; h.fhead = a2.w;
		e.L = p64(a3.x, a3.y); e.U = p64(a3.z, a3.w); /* (= the tail of the current list in an exact tail) */
		nx.tL = p64(a4.x, a4.y); nx.tU = p64(a4.z, a4.w);
		{ const uint4 a14 = mysave[14]; nx.fL = p64(a14.x, a14.y); nx.fU = p64(a14.z, a14.w); }
		nxi.L = p64(a5.x, a5.y); nxi.U = p64(a5.z, a5.w);
		e.f = a6.x; e.sa = a6.y; e.runsLo = a6.z; e.runsHi = a6.w;
		h.pused = a7.x; h.xhead = a7.y; e.runs2Lo = a7.z; e.runs2Hi = a7.w;
		h.neW = ((uint64_t)a8.y << 32) | a8.x; h.cb = (int)a8.z; h.cst = a8.w;
		h.side_load(); /* (the side buckets' states went to memory when the read was parked) */
		h.num_entries = (int)a9.x; r_vis_s = a9.y; r_vis_a = a9.z; r_pop = a9.w;
		r_push = a10.x;
		h.tw = v4(a11); h.tw1 = v4(a12); h.sw = v4(a13);
		if (!WIDE) e.runsLo = e.runsHi = e.runs2Lo = e.runs2Hi = ~0u; /* (16-byte entries: the one gap run is part of e.sa, LHeap::unpack) */
		rec.w = 0xFFFF0000u;
		active = true;
		__hip_atomic_fetch_add((Lds<unsigned int>)&s_active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	} else {
		for (int k = 0; k < nb; k++) h.bstate[k] = NONE32;
	}
	__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): nothing is pending when the loop is entered (what a resume loads is in its registers) - see the end of the loop */

	for (;;) {
		STAMP(7);
		bool admit = !active && !done;
		if (admit) {
			/* admission: what a read will need is not known in advance, and a read that finds the pool empty is given up and
			 * re-run, which costs far more than waiting.  So once three quarters of the region's shared part are taken a block starts a
			 * read only against memory it can see: its recycle stack plus its share of what is left of the region. */
			const uint32_t used = h.pshared + __hip_atomic_load(h.pool_bump, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (used >= h.pool_cap - ((h.pool_cap - h.pshared) >> 2)) {
				const uint32_t left = used < h.pool_cap ? h.pool_cap - used : 0u;
				const uint32_t blocks_in_region = (gridDim.x + sc.n_regions - 1) / sc.n_regions;
				const uint32_t avail = s_nfree + left / blocks_in_region;
				const uint32_t mine = __popcll(wballot(true) & ((1ull << lane) - 1ull));
				/* what a read is expected to take: four times the mean of the reads this block has finished (the needs are heavy-tailed;
				 * 150 bp reads with -n 5 average 1 750 chunks at GRCh37 scale, 100 bp reads with -n 3 a third of that), at least 1 MB */
				const uint32_t cnt = s_need_cnt, per = cnt ? (uint32_t)(((unsigned long long)s_need_sum * 64ull) / cnt) : 0u;
				const uint32_t want = per > ADMIT_CHUNKS ? per : ADMIT_CHUNKS;
				admit = avail / want >= mine + 1 || s_active + mine < 4;
			}
		}
		bool park = false;
		if (admit) {
			{ const uint32_t w_ = atomicAdd(R_wk(counter), 1u); /* (grab_read; the compiler folds this into one atomic per wave) */
			  rid = w_ >= R_wk(n_work) ? NONE32 : (R_wk(worklist) ? R_wk(worklist)[w_] : w_); }
			if (rid == NONE32) { if (R_wk(suspend)) park = true; else done = true; }
			else {
				const Batch &b = R_descs[R_wk(slot)].b;
				SET_MYSLOT(R_wk(slot));
				__hip_atomic_fetch_add((Lds<unsigned int>)&s_active, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				r_vis_s = r_vis_a = r_pop = r_push = 0;
				const uint32_t len_in = b.lens[rid];
				const bool unrep = len_in == BAD_LEN;
				SET_LEN(unrep ? 0u : len_in);
				const uint8_t *seq = b.reads + (size_t)rid * b.stride;
				rec.w = 0xFFFF0000u; /* (no record of this read yet) */
				const int cntN = b.dbuf[(size_t)rid * b.dstride + b.dstride - 4];
				h.reset(); /* heap_reset :540-546 (bucket states were cleared when the previous read finished) */
				n_alns = 0; exact_mode = false; active = true;
				bool ovf0 = false;
				/* a read whose calculate_d overflowed its scratch class waits for the re-run of both kernels in a larger class */
				const bool dfail = b.status[rid] == ST_D_OVF;
				/* (an EMPTY read is searched like any other: its root entry is a hit with the whole index as its interval, :331-344) */
				bool skip = cntN > kq.max_diff || unrep || dfail; /* inexact_match.c:260-266 */
				seeding = false;
				if (kq.use_precalc && !skip) {
					/* -P.  A read with an N in the last 12 bases of rc (= the first 12 of seq) gets an empty record
					 * (inexact_match.c:129-136).  Otherwise the heap starts from the precalculated list of that 12-mer
					 * (:269-279); the list is exact_match() of the 12-mer (align.c:212-216), a pure function of it, so the
					 * 12 exact steps are run here instead of reading the 16.7 M-list .pre table. */
					for (int k = 0; k < PRECALC_LEN; k++) if (seq[k] > 3) skip = true;
					if (!skip) {
						seeding = true;
						cL = 0; cU = last_row; curT = 1; cursel = false; s = 0; r = rd_len - 1;
						nx.T = 0; nxw = 0;
						exact_mode = true;
					}
				}
				if (!skip && !seeding) {
					/* heap_push(root) inexact_match.c:281 */
					h.cst = h.template reserve<true>(NONE32, 1, ovf0, xs);
					if (!ovf0) { h.cst++; h.set_top((P)0, last_row, (uint32_t)rd_len, 0u, ~0u, ~0u); h.store_packed(h.cst, h.tw, h.tw1); h.cprev = NONE32; h.mark(0); h.num_entries = 1; }
					r_push++;
				}
				SET_BEST_SCORE(kq.num_buckets); /* aln_score(max_diff+1,max_gapo+1,max_gape+1) :284 */
				SET_MAX_DIFF(kq.max_diff); num_best = 0;
				if (ovf0 || dfail) {
					if (ovf0) b.status[rid] = ST_SCRATCH_OVF;
					R_descs[R_wk(slot)].out.n[rid] = 0; (void)h.release_excess(xs); active = false;
					__hip_atomic_fetch_add((Lds<unsigned int>)&s_active, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					publish_done(R_descs[R_wk(slot)].done);
				}
			}
		}
		if (wk.slice_iters && w_iter >= wk.slice_iters) park = true;
		if (wany(park)) {
			/* ---- end of the slice for this wave: park the reads under way (word 0 of the save area tells the next launch) ---- */
			if (active) {
				__hip_atomic_fetch_add((Lds<unsigned int>)&s_active, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); /* (lanes of other waves may be waiting for admission) */
				auto lo = [](P v) { return (uint32_t)v; };
				auto hi = [](P v) { return (uint32_t)((uint64_t)v >> 32); };
				auto u4 = [](const u32x4 a) { return make_uint4(a.x, a.y, a.z, a.w); };
				const uint32_t fl = 1u | ((exact_mode ? 1u : 0u) << 1) | ((cursel ? 1u : 0u) << 2) | ((seeding ? 1u : 0u) << 3) | ((nxi_valid ? 1u : 0u) << 4) |
				                    ((h.top_valid ? 1u : 0u) << 5) | ((h.sec_valid ? 1u : 0u) << 6);
				mysave[0] = make_uint4(fl, rid, rdw, (uint32_t)num_best);
				mysave[1] = make_uint4((uint32_t)n_alns, (uint32_t)r, (uint32_t)s, (uint32_t)curT);
				mysave[2] = make_uint4((uint32_t)nx.T, nxw, h.cprev, h.fhead);
				mysave[3] = make_uint4(lo(e.L), hi(e.L), lo(e.U), hi(e.U));
				mysave[4] = make_uint4(lo(nx.tL), hi(nx.tL), lo(nx.tU), hi(nx.tU));
				mysave[14] = make_uint4(lo(nx.fL), hi(nx.fL), lo(nx.fU), hi(nx.fU));
				mysave[5] = make_uint4(lo(nxi.L), hi(nxi.L), lo(nxi.U), hi(nxi.U));
				mysave[6] = make_uint4(e.f, e.sa, e.runsLo, e.runsHi);
				mysave[7] = make_uint4(h.pused, h.xhead, e.runs2Lo, e.runs2Hi);
				mysave[8] = make_uint4((uint32_t)h.neW, (uint32_t)(h.neW >> 32), (uint32_t)h.cb, h.cst);
				mysave[9] = make_uint4((uint32_t)h.num_entries, r_vis_s, r_vis_a, r_pop);
				mysave[10] = make_uint4(r_push, 0u, 0u, 0u);
				mysave[11] = u4(h.tw); mysave[12] = u4(WIDE ? h.tw1 : h.tw); mysave[13] = u4(h.sw);
				h.side_flush();
			}
			parked = active;
			break;
		}
		if (wall(done)) break;
		if (!wany(active)) __builtin_amdgcn_s_sleep(64); /* a wave whose lanes all wait for admission */

		bool finish = false, ovf = false, from_pop = false, need_rank = false, alpha = false, is_group = false;
#ifdef BWB_PERTURB_VALU /* measurement only: what does the loop pay for 128 more vector instructions per iteration? (profiles/r4_ab_steps.txt) */
		asm volatile(".rept 128\n\tv_nop\n\t.endr");
#endif
#ifdef BWB_PERTURB_SALU
		asm volatile(".rept 128\n\ts_nop 0\n\t.endr");
#endif
		uint32_t ld_cnt = 0; /* heap entries this lane fetches from memory in this iteration */
		uint32_t pf_top = NONE32, pf_hdr = NONE32, pf_free = NONE32; /* what LHeap::pop wants fetched ahead of the gather; the chunk it emptied */
		P iL = 0, iU = 0;
		int widx = 0;
		n_iter = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_iter + (uint32_t)__popcll(wballot(active)))); /* (wave-uniform, like w_iter and n_bkt: lane 0 reports them) */
		w_iter = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w_iter + 1u));
		HIST(H_ITER, active); HISTW(H_WAVE_ITERS, 1);
#ifdef BWB_HIST
		int hw_g = 0, hw_x = 0, hw_0 = 0; bool h_mirror = false;
#endif

		/* add_alignment (align.c:271-298) into the lane's private hit list */
		auto add_aln = [&](P L, P U, int score, int alen) {
			const int e_go = (e.f >> 16) & 255;
			if (e_go) {
				for (int j = 0; j < n_alns; j++) {
					const uint4 a = myalns[j * ALN_U4];
					if (a.x == (uint32_t)L && a.y == (uint32_t)((uint64_t)L >> 32) && a.z == (uint32_t)U && a.w == (uint32_t)((uint64_t)U >> 32)) return;
				}
			}
			if (n_alns >= (int)sc_acap) { ovf = true; return; }
			myalns[n_alns * ALN_U4] = make_uint4((uint32_t)L, (uint32_t)((uint64_t)L >> 32), (uint32_t)U, (uint32_t)((uint64_t)U >> 32));
			myalns[n_alns * ALN_U4 + 1] = make_uint4((uint32_t)(score & 0xFFFF) | ((e.f << 8) & 0xFFFF0000u), (e.f >> 24) | ((uint32_t)(alen & 255) << 16), ERUNS_LO, ERUNS_HI); /* bwb_aln: score16 | mm | go, ge | - | alen16 */
			myalns[n_alns * ALN_U4 + 2] = make_uint4(ERUNS2_LO, ERUNS2_HI, 0u, 0u); /* gap runs 4..7 */
			n_alns++;
		};

		STAMP(0);
		/* ---- A: pick the SA interval of this iteration ---- */
		/* (Flat on purpose: with 64 reads per wave every path below is taken by some lane in nearly every iteration - tools/bbprof.py counts
		 * 0.99 executions per wave iteration for all of them - so a nest of branches buys nothing and costs, per level, the scalar mask
		 * bookkeeping and the copies of every value the branches merge: rounds 3-4 had four levels here.) */
		const bool ex = active && exact_mode;      /* exact_match_bounded exact_match.c:82-115: interval s of the current list, read char rc[r] */
		const bool popping = active && !exact_mode;
		const bool may_pop = popping && !(h.num_entries == 0 || h.num_entries > kq.max_entries); /* :293,299 */
		/* (count-only entries, see the expansion: when nothing is STORED any more, what the reference pops here is an entry whose score is above
		 * best_score + mm_score - it stops, :309-311 - and which this heap never held) */
		const bool can_pop = may_pop && (!ph_ok || h.neW != 0ull);
		if (can_pop) {
			if (!(h.neW & 1ull)) h.switch_cache(h.best(nb)); /* (bit 0 of the window = the cached bucket: still non-empty in all but a few dozen pops per read) */
#ifdef BWB_HIST
			h_mirror = h.top_valid;
#endif
			h.pop(e, ld_cnt, pf_top, pf_hdr, pf_free); /* heap_pop :594-610: the top of the best bucket from its register mirror; what it uncovers is fetched now, under the gather */
		}
		{
			/* a deletion group is not an entry of the reference's heap: the pop that the reference makes here is that of the group's last
			 * child, which happens in the next iteration, once the children are in place */
			const bool grp = can_pop && (e.sa & 3u) == (uint32_t)STATE_GROUP;
			/* :309 (the reference pops that child, then stops).  aln_entry_t.score is an 8-bit field (align.h:104): what the reference compares is
			 * the score modulo 256 - the same number unless the parameters allow scores above 255 */
			const bool over = can_pop && (e_score & 255) > rd_best_score + h.pXv;
			finish = popping && (!can_pop || over);
			r_pop += ((can_pop && (!grp || over)) || (may_pop && !can_pop)) ? 1u : 0u;
			HIST(H_PH_POP, may_pop && !can_pop);
			from_pop = can_pop && !over;
			is_group = grp && !over;                   /* the children are those of the parent's O_alphabet call (:382-383) */
			h.num_entries += is_group ? 1 : 0;
			widx = ex ? r + 1 : (int)(e.f & 255u);
			need_rank = ex || (from_pop && (is_group || widx > 0));
			/* an entry with no difference left goes to the exact tail (exact counts); any other one is expanded with O_alphabet (:345,382) */
			alpha = MULTI && from_pop && (is_group || (rd_max_diff - (int)((e.f >> 8) & 255u) - (int)((e.f >> 16) & 255u) - (int)(e.f >> 24)) != 0);
			iL = e.L; iU = e.U;                        /* (the popped entry's interval - and, in an exact tail, the tail of the current list: cL / cU) */
		}
		{
			const bool mid = ex & (s != curT - 1), take = mid & nxi_valid;
			iL = take ? nxi.L : iL; iU = take ? nxi.U : iU; /* fetched at the end of the previous step */
			if (mid & !nxi_valid) { /* (only the iteration after a resume: waited for inside the branch, see LHeap::pop) */
				const Intv<P> v = (lbase + (cursel ? lcap : 0))[s]; iL = v.L; iU = v.U;
				asm volatile("" :: "v"(iL), "v"(iU));
			}
		}
		/* the interval of the step's NEXT iteration, when it is one of the list in memory (not its tail, which is in registers): fetched
		 * now, under this iteration's gather (round 3 fetched it at the end of the iteration and used it at the start of the next) */
		if (ex && s + 1 < curT - 1) nxi = (lbase + (cursel ? lcap : 0))[s + 1];
		h.prefetch(pf_top, pf_hdr); /* (the lanes have met again: see prefetch128) */

#ifdef BWB_HIST
		HIST(H_POP, active && !exact_mode && (from_pop || finish)); HIST(H_POP_GAPPED, from_pop && ((e.f >> 16) != 0)); HIST(H_POP_FROM_MIRROR, from_pop && h_mirror);
		HIST(H_EXACT_STEP, active && exact_mode); HIST(H_NEED_RANK, need_rank); HIST(H_ALPHA, need_rank && alpha);
		HIST(H_EXACT_MULTI, active && exact_mode && curT > 1); HIST(H_ALLOC, from_pop && is_group);
		const P hw_ = (P)(iU - iL + 1);
		const bool hsame_ = need_rank && ((P)(iL - 1) >> 7) == (iU >> 7) && iL != 0 && iU != last_row;
		HIST(H_SAME_BKT, hsame_); HIST(H_TWO_BKT, need_rank && !hsame_);
		HIST(H_W1, need_rank && hw_ == 1); HIST(H_W2, need_rank && hw_ == 2); HIST(H_W4, need_rank && hw_ > 2 && hw_ <= 4); HIST(H_W8, need_rank && hw_ > 4 && hw_ <= 8);
		HIST(H_W32, need_rank && hw_ > 8 && hw_ <= 32); HIST(H_W128, need_rank && hw_ > 32 && hw_ <= 128); HIST(H_WBIG, need_rank && (hw_ > 128 || hw_ == 0));
		HISTW(H_WAVE_ANY_TWO_BKT, wany(need_rank && !hsame_) ? 1 : 0); HISTW(H_WAVE_ANY_WIDE8, wany(need_rank && (hw_ > 8 || hw_ == 0)) ? 1 : 0);
		HISTW(H_WAVE_ANY_EXACT, wany(active && exact_mode) ? 1 : 0); HISTW(H_WAVE_ANY_EXPAND, wany(from_pop && need_rank) ? 1 : 0);
		HISTW(H_WAVE_ALL_EXACT, wall(!active || exact_mode) ? 1 : 0);
#endif
		STAMP(1);
		/* ---- B: one round of memory: D words, read base, both rank buckets and the side heap buckets, issued together ---- */
		uint32_t wd = 0, ws = 0, ne = 0;
		int cr = 4;
		const unsigned long long rmask = wballot(need_rank);
		const int nreq = __popcll(rmask);
		const bool want_rec = need_rank || (from_pop && rd_len < kq.seed_length); /* (a finished entry, i == 0, of a read shorter than the seed still meets the seed bound, :324-328) */
		/* The record of the entry's position (D[i-1], D[i-2] | D_seed pair | seq[len - widx]: bwb_kernels.h) - the one in registers when its tag
		 * matches, else one load.  It is only ISSUED here, in place (prefetch128), and unpacked after the rank: round 3 unpacked it on the spot
		 * - a flat load, which the gather's wait for its exchange array waits for as well - so a wave sat out the record's round trip before
		 * its gather was even issued, a round trip that the gather's own wait covers for free. */
		const bool rec_load = want_rec && (rec.w >> 16) != (uint32_t)(widx >> 2);
		{
			const unsigned long long mr = wballot(rec_load);
			n_rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(n_rec + (uint32_t)__popcll(mr)));
			{ /* the read's records written by kl_calc_d: 16 bytes per four positions (bwb_kernels.h: rec_put); unconditional like LHeap::prefetch */
				const uint32_t sl_ = rd_myslot;
				const unsigned char *rb = (const unsigned char *)(uintptr_t)((Lds<unsigned long long>)&s_dbuf[0])[sl_] + (size_t)rid * ((Lds<unsigned int>)&s_dstride[0])[sl_];
				prefetch128(rec, rb + REC_BYTES * (widx >> 2), mr);
			}
		}
		STAMP(2);
		HISTW(H_WAVE_NREQ_LE16, nreq <= 16 ? 1 : 0);
		KidCtx<P> kc;
		kc.R = (uint32_t)(uintptr_t)stage; kc.baseL = kc.baseU = sb; kc.qL = kc.qU = false; kc.nvis = 0;
		if (nreq > 0) ne = wave_children<P>(buckets, last_row, need_rank, iL, iU, alpha, sb, stage, zero_row, lane, n_bkt, kc); /* every lane of the wave loads */
		__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): what was issued ahead of the gather (prefetch128 / prefetch32: the record, LHeap::prefetch) has landed, also when no lane needed a rank */
		h.give_back(pf_free);
		{ /* (unconditional: a lane that wants no record reads none of the three - and a region around eleven instructions costs four) */
			/* bytes j, j + 1 of a record's six D bytes are D[i-2], D[i-1] for i = 4 m + j; wd / ws hold them as {D[i-1] low, D[i-2] high} */
			/* (one v_perm_b32 per pair: byte j + 1 of the eight bytes {rec.y : rec.x} into byte 0, byte j into byte 1, zeros above) */
			const uint32_t jb = (uint32_t)widx & 3u;
			const uint32_t sel = jb * 0x101u + 0x0c0c0001u;
			wd = __builtin_amdgcn_perm(rec.y, rec.x, sel); ws = __builtin_amdgcn_perm(rec.w, rec.z, sel);
			const int cf = (int)((rec.y >> (16u + 4u * jb)) & 15u); /* rc[widx-1] = complement of seq[len-1-(widx-1)] (io.c:502-504) */
			cr = cf > 3 ? 4 : 3 - cf;
		}
		STAMP(14);
		/* heap buckets an expansion of this entry can push to besides its own: mismatch, gap (:434-504); their states are in registers (LHeap) */
		const int e_state = (int)(e.sa & 3u);
		const int scX = e_score + h.pXv, scG = e_score + (e_state == STATE_M ? h.pGov : h.pGev);
		const int wG = e_state == STATE_M ? h.side_of(kq.gapo_score) : h.side_of(kq.gape_score); /* which register is the gap bucket's (0: the cached bucket itself) */
		/* -S (O_actg_alphabet bwt.c:440-463): only A, G, C, T exist as children, pushed in that order (:434-504 with alphabet_size 5):
		 * child rows 1..4 stand for the codes 15, 3, 7, 1 */
		if (!MULTI) ne = (((ne >> 15) & 1u) << 1) | (((ne >> 3) & 1u) << 2) | (((ne >> 7) & 1u) << 3) | (((ne >> 1) & 1u) << 4);
		uint32_t st_cnt = 0; /* heap entries this lane stores in this iteration */
		auto kid = [&](int j, P &L, P &U) { kid_get<P>(kc, sb, MULTI ? j : (int)((0x173Fu >> (4 * (j - 1))) & 15u), L, U); };
		STAMP(3);

		/* ---- C: act on it ---- */
		bool exact_step = active && (exact_mode);
		if (from_pop && is_group) {
			/* ---- a deletion group has reached the top of its bucket: its children take its place ---- */
			/* (a COMBINED group - the run it opens has its deletion bit clear - also holds the insertion that the reference pushed in front of the
			 * deletions, :440-443: that entry goes in first, below the deletion children) */
			uint64_t gr = ((uint64_t)ERUNS_HI << 32) | ERUNS_LO, gr2 = ((uint64_t)ERUNS2_HI << 32) | ERUNS2_LO;
			const int g_run = (int)((e.f >> 16) & 255u) - 1;                 /* the run the group's gap belongs to: the children's num_gapo - 1 */
			const int g_sh = 16 * (g_run & 3) + 15;
			const bool g_hi = WIDE && g_run >= 4;
#ifdef BWB_NO_GAP_COMBINE
			const bool comb = false;
#else
			const bool comb = (((g_hi ? gr2 : gr) >> g_sh) & 1ull) == 0ull;
#endif
			const int n = __popc(ne) + (comb ? 1 : 0);
			const uint32_t cst_old = h.cst;
			const uint32_t st0 = h.reserve(cst_old, n, ovf, xs);
			if (!ovf && n > 0) {
				if (st0 != cst_old) h.cprev = cst_old; /* (a new chunk was started: its header names the state before it) */
				const uint32_t sd = (e.sa & 0x3FCu) | (uint32_t)STATE_D; /* (aln_length; the runs travel apart) */
				uint32_t sx = st0, gm = ne;
				if (comb) {
					u32x4 w0, w1;
					h.pack(e.L, e.U, (e.f & ~255u) | ((e.f - 1u) & 255u), (e.sa & 0x3FCu) | (uint32_t)STATE_I, (uint32_t)gr, (uint32_t)(gr >> 32), w0, w1, (uint32_t)gr2, (uint32_t)(gr2 >> 32));
					h.store_packed(++sx, w0, w1); st_cnt++;
					if (!WIDE) h.sw = w0;
					if (g_hi) gr2 |= 1ull << g_sh; else gr |= 1ull << g_sh; /* the deletions' version of the run */
				}
				while (gm) {
					const int j = __ffs((int)gm) - 1;
					gm &= gm - 1;
					P cl, cu;
					kid(j, cl, cu);
					u32x4 w0, w1;
					h.pack(cl, cu, e.f, sd, (uint32_t)gr, (uint32_t)(gr >> 32), w0, w1, (uint32_t)gr2, (uint32_t)(gr2 >> 32));
					if (gm) { h.store_packed(++sx, w0, w1); st_cnt++; if (!WIDE) h.sw = w0; }
					else { h.tw = w0; h.tw1 = w1; } /* (the last child is popped next: the register mirror is its only copy) */
				}
				h.top_valid = true; h.sec_valid = !WIDE && n >= 2; /* (n == 1 cannot happen: a single deletion child without an insertion is stored as itself) */
				h.cst = st0 + (uint32_t)n; h.mark(e_score);
			}
		}
		/* (the four things a popped entry can be are siblings, not a nest: see section A) */
		const int e_i = e.f & 255, e_mm = (e.f >> 8) & 255, e_go = (e.f >> 16) & 255, e_ge = (e.f >> 24) & 255;
		const int e_alen = (int)((e.sa >> 2) & 255u);
		const int diff_left = rd_max_diff - e_mm - e_go - e_ge;
		const int diff_left_seed = kq.max_diff_seed - e_mm - e_go - e_ge;
		const int seed_index = e_i - (rd_len - kq.seed_length);
		/* (`|` and `&` on purpose, here and below: with `||` / `&&` the compiler builds a divergent region per short circuit - mask, branch,
		 * restore - around three or four instructions that every wave executes anyway: tools/bbprof.py counted 82 such regions per iteration) */
		const bool pruned = (diff_left < 0)                                                        /* :313 */
		                    | ((e_i > 0) & (diff_left < (int)(wd & 127u)))                          /* :317 */
		                    | ((seed_index > 0) & (diff_left_seed < (int)(ws & 127u)));             /* :326 */
		const bool live = from_pop && !is_group && !pruned;
		const bool do_hit = live && e_i == 0, do_tail = live && e_i != 0 && diff_left == 0, do_expand = live && e_i != 0 && diff_left != 0;
#ifdef BWB_HIST
		HIST(H_PRUNED, from_pop && !is_group && pruned); HIST(H_HIT, do_hit); HIST(H_EXACT_START, do_tail);
		HIST(H_EXPAND, do_expand);
		HIST(H_EXP_SAME_W1, do_expand && hsame_ && hw_ == 1); HIST(H_EXP_SAME_W2_4, do_expand && hsame_ && hw_ >= 2 && hw_ <= 4);
#endif
		if (do_hit) { /* hit :331-344 */
			if (n_alns == 0) {
				SET_BEST_SCORE(e_score);
				const int bd = e_mm + e_go + e_ge;
				SET_MAX_DIFF((bd + 1 > kq.max_diff) ? kq.max_diff : bd + 1);
			}
			if (e_score == rd_best_score) { num_best += (int)(uint32_t)(e.U - e.L + 1); add_aln(e.L, e.U, e_score, e_alen); }
			else if (num_best > kq.max_best) finish = true;
			else add_aln(e.L, e.U, e_score, e_alen);
		}
		{ /* exact tail :345-375: its first step uses the children just computed (cL / cU are e.L / e.U already) */
			curT = do_tail ? 1 : curT; cursel = do_tail ? false : cursel; s = do_tail ? 0 : s; r = do_tail ? e_i - 1 : r;
			nx.T = do_tail ? 0 : nx.T; nxw = do_tail ? 0u : nxw;
			exact_mode = exact_mode || do_tail;
			exact_step = exact_step || do_tail;
		}
		if (do_expand) {
			STAMP(8);
			/* ---- expansion :377-504 ---- */
			r_vis_a += (uint32_t)kc.nvis;
			/* :386-432, as expressions (a mismatch that the bound forbids does not matter once no difference is allowed at all) */
			const int d1 = wd & 255u, d2 = (wd >> 8) & 255u, d1s = ws & 255u, d2s = (ws >> 8) & 255u;
			const bool cD = e_i - 1 > 0, cS = seed_index - 1 > 0;
			const bool allow_diff = !((cD & ((diff_left - 1) < (d2 & 127))) | (cS & ((diff_left_seed - 1) < (d2s & 127))));
			const bool allow_mm = !((cD & ((d1 & 127) == diff_left - 1) & ((d2 & 127) == diff_left - 1) & ((d1 & 128) != 0))
			                        | (cS & ((d1s & 127) == diff_left_seed - 1) & ((d2s & 127) == diff_left_seed - 1) & ((d1s & 128) != 0)));
			const int tmp = e_go + e_ge;
			const bool allow_open = e_go < kq.max_gapo, allow_extend = e_ge < kq.max_gape;
			const bool allow_indels = !((e_i - 1 < kq.no_indel_length + tmp) | ((rd_len - (e_i - 1)) < kq.no_indel_length + tmp)) & (allow_open | allow_extend);
			const bool gap_open = e_state == STATE_M;
			const int sc0 = e_score;
			const bool ins_ok = allow_diff & allow_indels & (((e_state == STATE_I) & allow_extend) | ((e_state == STATE_M) & allow_open));
			const bool del_ok = allow_diff & allow_indels & (e_state != STATE_I) & (e_state == STATE_M ? allow_open : allow_extend);
			const bool mm_ok = allow_diff & allow_mm;
			const uint32_t mem = cr > 3 ? 0u : (MULTI ? member_mask(cr) : 2u << cr);
			/* push sequence (:434-504): insertion, deletions j = 1..15, then match/mismatch j = 1..15 */
			const uint32_t delm = del_ok ? ne : 0u;
			const uint32_t mgrp = mm_ok ? ne : (ne & mem);
			const uint32_t matchm = mgrp & mem, mism_c = mgrp & ~mem;
			const int nDel_c = __popc(delm), nIns_c = ins_ok ? 1 : 0, nX_c = __popc(mism_c), n0 = __popc(matchm);
			const int nGc = nIns_c + nDel_c;        /* gap entries the reference pushes: what is counted */
			const int nPush = nGc + nX_c + n0;
			r_push += nPush;
			/* COUNT-ONLY PUSHES (round 6).  Once the read has its first hit best_score is fixed (:333-337,349-353: set only while the read has no
			 * alignment); entries are popped in non-decreasing score order (:594-610) and the loop stops at the first popped entry whose score is
			 * above best_score + mm_score (:309-311) - exactly as it stops on an empty heap (:293).  A child pushed with such a score can therefore
			 * influence the result only through heap->num_entries (:299-301): it is counted (num_entries, pushes) and NOT stored - no slot, no chunk,
			 * no bucket mark.  Before the first hit best_score is the number of buckets and no child qualifies.  With the default penalties that is
			 * every gap child and, for a parent above best_score, every mismatch child pushed after the first hit: more than half of all entries
			 * the round-5 kernel stored, and nearly all that stayed in the pool until the end of the read.  (ph_ok: the scores compared are the true
			 * ones only while they fit the reference's 8-bit field, align.h:104; and the window must see every stored bucket.) */
			const bool ph_g = ph_ok & (scG > rd_best_score + h.pXv), ph_x = ph_ok & (e_score > rd_best_score);
			const int nDel = ph_g ? 0 : nDel_c, nIns = ph_g ? 0 : nIns_c;
			const uint32_t mism = ph_x ? 0u : mism_c;
			const int nX = ph_x ? 0 : nX_c;
#ifdef BWB_NO_GAP_COMBINE /* (A/B: the insertion as its own entry next to the deletion group = rounds 3-5) */
			const int nG = nIns + (nDel ? 1 : 0);   /* gap entries stored: the deletions as one group (STATE_GROUP) */
#else
			/* gap entries stored: ONE.  An expansion that opens gaps pushes the insertion and then the deletions onto the same bucket (:438-463;
			 * both or neither: `ins_ok` and `del_ok` are the same condition for a STATE_M parent), and all of them derive from the parent's
			 * interval - so the insertion rides in the deletion group (round 6: a COMBINED group, told by the clear deletion bit of the run it
			 * opens) and is put in place, below the deletion children, when the group reaches the top of its bucket.  Gap entries were two
			 * thirds of all entries the round-5 kernel stored at GRCh37 scale, almost none of them ever popped. */
			const int nG = (nIns | nDel) ? 1 : 0;
#endif
#ifdef BWB_HIST
			{ const int nne = __popc(ne);
			  HIST(H_NE0, nne == 0); HIST(H_NE1, nne == 1); HIST(H_NE2, nne == 2); HIST(H_NE3_4, nne == 3 || nne == 4); HIST(H_NE5_8, nne >= 5 && nne <= 8); HIST(H_NE9, nne >= 9);
			  HIST(H_DEL_OK, del_ok); HIST(H_MM_OK, mm_ok); HIST(H_INS_OK, ins_ok);
			  hw_g = nG; hw_x = nX; hw_0 = n0; hl_gap += nGc; hl_mis += nX_c; hl_match += n0;
			  hl_phg += ph_g ? nGc : 0; hl_phx += ph_x ? nX_c : 0; HIST(H_EXPAND_AFTER_HIT, n_alns != 0); }
#endif
			/* target buckets: 0 = sc0 (the cached one), 1 = scX, 2 = scG; equal scores share a bucket in sequence order */
			const int tX = kq.mm_score == 0 ? 0 : 1, tG = wG == 0 ? 0 : (wG == 1 ? 1 : 2);
			const int k0 = n0 + (tX == 0 ? nX : 0) + (tG == 0 ? nG : 0);
			const int k1 = (tX == 1 ? nX : 0) + (tG == 1 ? nG : 0);
			const int k2 = tG == 2 ? nG : 0;
			STAMP(9);
			const uint32_t cst_old = h.cst;
			uint32_t st0 = h.reserve(cst_old, k0, ovf, xs);
			const uint32_t stX = h.stX, vGo = h.stGo, vGe = h.stGe, stG = wG == 2 ? vGo : vGe;
			uint32_t st1 = h.reserve(stX, k1, ovf, xs);
			uint32_t st2 = h.reserve(stG, k2, ovf, xs);
			STAMP(10);
			if (!ovf) {
				/* child entry templates */
				const uint32_t alen1 = (uint32_t)((e_alen + 1) & 255);
				const uint32_t f_base = ((uint32_t)e_go << 16) | ((uint32_t)e_ge << 24);
				const uint32_t f_match = (uint32_t)((e_i - 1) & 255) | ((uint32_t)e_mm << 8) | f_base;
				const uint32_t f_mis = (uint32_t)((e_i - 1) & 255) | ((uint32_t)((e_mm + 1) & 255) << 8) | f_base;
				const uint32_t f_gap = ((uint32_t)e_mm << 8) | ((uint32_t)((e_go + (gap_open ? 1 : 0)) & 255) << 16) | ((uint32_t)((e_ge + (gap_open ? 0 : 1)) & 255) << 24);
				const uint64_t eruns = ((uint64_t)ERUNS_HI << 32) | ERUNS_LO, eruns2 = ((uint64_t)ERUNS2_HI << 32) | ERUNS2_LO;
				uint64_t gruns_i, gruns_d, gruns2 = eruns2; /* new run on open (start = aln_length, len 1); len+1 on extend; runs 4..7 (32-byte entries) in gruns2 */
				if (!WIDE) { /* one run at most (max_gapo <= 1: the run opened is run 0, the run extended is run 0): 32-bit arithmetic */
					const uint32_t run0 = (e.sa >> 10) & 0xFFFFu;
					gruns_i = gap_open ? ((uint32_t)e_alen | 0x100u) : run0 + 0x100u;
					gruns_d = gap_open ? ((uint32_t)e_alen | 0x8100u) : run0 + 0x100u;
				} else if (gap_open) {
					const int sh = 16 * (e_go & 3);
					const bool hi4 = e_go >= 4; /* (a fifth to eighth gap open: the run goes into the second word pair) */
					const uint64_t src = hi4 ? eruns2 : eruns;
					const uint64_t cleared = src & ~(0xFFFFull << sh);
					const uint64_t vi = cleared | ((uint64_t)((uint32_t)e_alen | 0x100u) << sh), vd = cleared | ((uint64_t)((uint32_t)e_alen | 0x8100u) << sh);
					gruns_i = hi4 ? eruns : vi; gruns_d = hi4 ? eruns : vd;
					if (hi4) gruns2 = vi; /* (the deletion's version: bit 15 of the new run, gruns2_d below) */
				} else {
					const bool hi4 = e_go - 1 >= 4;
					const uint64_t inc = 0x100ull << (16 * ((e_go - 1) & 3));
					gruns_i = gruns_d = hi4 ? eruns : eruns + inc;
					if (hi4) gruns2 = eruns2 + inc;
				}
				/* the deletion's version of the second word pair: the run just opened there carries bit 15 */
				const uint64_t gruns2_d = (WIDE && gap_open && e_go >= 4) ? (gruns2 | (0x8000ull << (16 * (e_go & 3)))) : gruns2;
				/* the slot last used on every target bucket (a state word is also the slot's index in the pool: chunk << 6 | fill) */
				uint32_t s0 = st0, s1 = st1, s2 = st2;
				auto emit = [&](uint32_t &sx, P L, P U, uint32_t f, uint32_t sa, uint64_t runs, uint64_t runs2) {
					u32x4 w0, w1;
					h.pack(L, U, f, sa, (uint32_t)runs, (uint32_t)(runs >> 32), w0, w1, (uint32_t)runs2, (uint32_t)(runs2 >> 32));
					h.store_packed(++sx, w0, w1);
					st_cnt++; /* (per lane and iteration; summed over the wave once per launch) */
				};
				bool top_ok = false; /* does the register mirror hold the last entry pushed on bucket sc0? */
				STAMP(11);
				{ /* gap pushes: insertion (keeps the interval), then the deletions of every non-empty code - as one group entry that
				   * holds the parent's interval (a single deletion child is stored as itself) */
					uint32_t sg = tG == 0 ? s0 : (tG == 1 ? s1 : s2);
					const uint32_t fd = f_gap | (uint32_t)(e_i & 255);
#ifdef BWB_NO_GAP_COMBINE
					if (nIns) emit(sg, e.L, e.U, f_gap | (uint32_t)((e_i - 1) & 255), (uint32_t)STATE_I | (alen1 << 2), gruns_i, gruns2);
					if (nDel == 1) { P cl, cu; kid(__ffs((int)delm) - 1, cl, cu); emit(sg, cl, cu, fd, (uint32_t)STATE_D | (alen1 << 2), gruns_d, gruns2_d); }
					else if (nDel) emit(sg, e.L, e.U, fd, (uint32_t)STATE_GROUP | (alen1 << 2), gruns_d, gruns2_d);
#else
					if (nIns && nDel) emit(sg, e.L, e.U, fd, (uint32_t)STATE_GROUP | (alen1 << 2), gruns_i, gruns2); /* combined: the INSERTION's runs (deletion bit clear) */
					else if (nIns) emit(sg, e.L, e.U, f_gap | (uint32_t)((e_i - 1) & 255), (uint32_t)STATE_I | (alen1 << 2), gruns_i, gruns2);
					else if (nDel == 1) { P cl, cu; kid(__ffs((int)delm) - 1, cl, cu); emit(sg, cl, cu, fd, (uint32_t)STATE_D | (alen1 << 2), gruns_d, gruns2_d); }
					else if (nDel) emit(sg, e.L, e.U, fd, (uint32_t)STATE_GROUP | (alen1 << 2), gruns_d, gruns2_d);
#endif
					if (tG == 0) s0 = sg; else if (tG == 1) s1 = sg; else s2 = sg;
				}
				STAMP(12);
				const uint32_t sm = (uint32_t)STATE_M | (alen1 << 2);
				if (kq.mm_score != 0) { /* mismatches and matches land on different buckets: two independent sequences */
					uint32_t sxm = tX == 1 ? s1 : s0;
					uint32_t xm = mism;
					while (xm) {
						const int j = __ffs((int)xm) - 1;
						xm &= xm - 1;
						P cl, cu;
						kid(j, cl, cu);
						emit(sxm, cl, cu, f_mis, sm, eruns, eruns2);
					}
					if (tX == 1) s1 = sxm; else s0 = sxm;
					/* The last match child is the next entry popped (same score, LIFO, and nothing is ever pushed below the bucket being
					 * popped): the register mirror is its only copy, its slot is reserved but never written.  The entry below it - the
					 * match child before it, or with a single child what the pop uncovered, when nothing else went on top of that -
					 * stays in the second mirror register. */
					/* (peeled, round 5: the top child and the one below it are straight-line code, the loop only runs for a third match child
					 * and beyond - no branch and no mirror copies inside a trip; slots are handed out in ascending code order all the same) */
					if (matchm) {
						const int jt = 31 - __clz((int)matchm);
						uint32_t mm = matchm & ~(1u << jt);
						if (mm) {
							const int j2 = 31 - __clz((int)mm);
							mm &= ~(1u << j2);
							while (mm) {
								const int j = __ffs((int)mm) - 1;
								mm &= mm - 1;
								P cl, cu;
								kid(j, cl, cu);
								u32x4 w0, w1;
								h.pack(cl, cu, f_match, sm, ERUNS_LO, ERUNS_HI, w0, w1, ERUNS2_LO, ERUNS2_HI);
								h.store_packed(++s0, w0, w1); st_cnt++;
							}
							P cl, cu;
							kid(j2, cl, cu);
							u32x4 w0, w1;
							h.pack(cl, cu, f_match, sm, ERUNS_LO, ERUNS_HI, w0, w1, ERUNS2_LO, ERUNS2_HI);
							h.store_packed(++s0, w0, w1); st_cnt++;
							if (!WIDE) { h.sw = w0; h.sec_valid = true; }
						} else if (!WIDE) { h.sw = h.tw; h.sec_valid = h.top_valid && k0 == 1; } /* (what the pop uncovered: from the second mirror register, or on its way from memory) */
						P cl, cu;
						kid(jt, cl, cu);
						h.pack(cl, cu, f_match, sm, ERUNS_LO, ERUNS_HI, h.tw, h.tw1, ERUNS2_LO, ERUNS2_HI);
						top_ok = true;
					}
				} else { /* mm_score == 0: one bucket, interleaved in code order */
					uint32_t am = matchm | mism;
					while (am) {
						const int j = __ffs((int)am) - 1;
						am &= am - 1;
						P cl, cu;
						kid(j, cl, cu);
						emit(s0, cl, cu, ((mem >> j) & 1u) ? f_match : f_mis, sm, eruns, eruns2);
					}
				}
				STAMP(13);
				h.num_entries += nPush;
				if (k0 > 0) { h.cst = st0 + (uint32_t)k0; h.mark(sc0); h.top_valid = top_ok; if (!top_ok) h.sec_valid = false; if (st0 != cst_old) h.cprev = cst_old; }
				if (k1 > 0) { h.stX = st1 + (uint32_t)k1; h.mark(scX); }
				if (k2 > 0) { const uint32_t v = st2 + (uint32_t)k2; if (wG == 2) h.stGo = v; else h.stGe = v; h.mark(scG); }
			}
		}

#ifdef BWB_HIST
		{ int t_ = hw_g; for (int o_ = 32; o_; o_ >>= 1) t_ = max(t_, __shfl_xor(t_, o_)); HISTW(H_WAVE_GAPLOOP, t_);
		  t_ = hw_x; for (int o_ = 32; o_; o_ >>= 1) t_ = max(t_, __shfl_xor(t_, o_)); HISTW(H_WAVE_MISLOOP, t_);
		  t_ = hw_0; for (int o_ = 32; o_; o_ >>= 1) t_ = max(t_, __shfl_xor(t_, o_)); HISTW(H_WAVE_MATCHLOOP, t_); }
#endif
		STAMP(4);
		if (exact_step && need_rank) {
			bool exact_done = false;
			uint32_t lastW = 0; /* summed width of the list that the step just completed */
			{ /* (selects, not a nest: one divergent region - the appends - per step) */
				const bool isN = cr > 3; /* N in the read: exact_match.c:84-87 */
				if (!isN && !seeding) r_vis_s += (uint32_t)kc.nvis; /* (the reference reads the list from its table) */
				uint32_t nm = isN ? 0u : (ne & (MULTI ? member_mask(cr) : 2u << cr));
				if (!isN && list_full<P>(nx, lcap)) { ovf = true; nm = 0; }
				while (nm) { /* ascending code order == nucl_bases_table order (io.h:102-106) */
					const int j = __ffs((int)nm) - 1;
					nm &= nm - 1;
					P cl, cu;
					kid(j, cl, cu);
					nxw += (uint32_t)(cu - cl + 1);
					list_add<P>(nx, lbase, cursel ? 0 : 1, cl, cu, lcap);
				}
				s += isN ? 0 : 1;
				const bool swap = !isN && !ovf && s >= curT;
				/* the list's swap; an N ends the tail with an empty list (curT = 0) */
				cursel = swap ? !cursel : cursel;
				cL = swap ? nx.tL : cL; cU = swap ? nx.tU : cU;
				const int newT = isN ? 0 : nx.T;
				const bool first2 = swap && newT >= 2;
				nxi.L = first2 ? nx.fL : nxi.L; nxi.U = first2 ? nx.fU : nxi.U; /* (the new list's first interval: from registers, not from what this step has just stored) */
				curT = (swap || isN) ? newT : curT;
				lastW = swap ? nxw : 0u;
				nx.T = swap ? 0 : nx.T; s = swap ? 0 : s; nxw = swap ? 0u : nxw;
				const bool more = swap && newT != 0; /* :114 */
				r -= more ? 1 : 0;
				exact_done = isN || (swap && (newT == 0 || r < (seeding ? rd_len - PRECALC_LEN : 0)));
				/* the interval of the next iteration, when it is not the list's tail (which is in registers): within a step it is on its way
				 * since the start of this iteration; the first interval of a NEW list was kept in registers by list_add (round 4's first
				 * version loaded what this step had just stored: the wait for it, at the start of the next iteration, was 10 % of the loop) */
				nxi_valid = !ovf && !exact_done && s != curT - 1;
			}
			STAMP(6);
			if (exact_done && !ovf && seeding) {
				/* :269-279: one entry per interval, i = readLen - 12, a 12-long all-M path; no interval: no alignment */
				exact_mode = false; seeding = false;
				if (curT == 0) finish = true;
				else {
					const uint32_t ent_f = (uint32_t)(rd_len - PRECALC_LEN), ent_sa = (uint32_t)STATE_M | ((uint32_t)PRECALC_LEN << 2);
					uint32_t st = h.cst; /* bucket 0 of the empty heap */
					for (int k = 0; k < curT && !ovf; k++) {
						P iL2, iU2;
						if (k == curT - 1) { iL2 = cL; iU2 = cU; }
						else { const Intv<P> v = (lbase + (cursel ? lcap : 0))[k]; iL2 = v.L; iU2 = v.U; }
						const uint32_t st_old = st;
						st = h.reserve(st_old, 1, ovf, xs);
						if (!ovf) { if (st != st_old) h.cprev = st_old; st++; h.set_top(iL2, iU2, ent_f, ent_sa, ~0u, ~0u); h.store_packed(st, h.tw, h.tw1); }
					}
					if (!ovf) { h.cst = st; h.mark(0); h.num_entries = curT; h.sec_valid = false; r_push += (uint32_t)curT; }
				}
			} else if (exact_done && !ovf) {
				exact_mode = false;
				if (curT != 0) { /* matches found :347-371 */
					const int e_i = e.f & 255, e_mm = (e.f >> 8) & 255, e_go = (e.f >> 16) & 255, e_ge = (e.f >> 24) & 255;
					if (n_alns == 0) {
						SET_BEST_SCORE(e_score);
						const int bd = e_mm + e_go + e_ge;
						SET_MAX_DIFF((bd + 1 > kq.max_diff) ? kq.max_diff : bd + 1);
					}
					bool brk = false;
					/* num_best += the width of every interval of the list (:350-352): adjacent intervals merge without changing the
					 * sum, so it is the running sum of what the last step added - no pass over the list in memory */
					if (e_score == rd_best_score) num_best += (int)lastW;
					else if (num_best > kq.max_best) brk = true;
					if (brk) finish = true;
					else {
						const int alen2 = ((int)((e.sa >> 2) & 255u) + e_i) & 255; /* :365 */
						const Intv<P> *lst = lbase + (cursel ? lcap : 0);
						/* the list's first interval (when it has several) is in registers - nxi, from the swap above - and so is its tail: a list
						 * of two costs no load at all; what lies in between is loaded several intervals at a time (round 3 loaded one per
						 * trip of a loop: a chain of dependent waits that, with some lane of a wave ending an exact tail in most iterations,
						 * took 9 % of the loop) */
						int k = 0;
						if (curT >= 2) { add_aln(nxi.L, nxi.U, e_score, alen2); k = 1; }
						if (e_go == 0) { /* no duplicate check (align.c:273-280 applies to gapped entries): four list loads in flight at a time */
							for (; k + 4 <= curT - 1 && n_alns + 4 <= (int)sc_acap; k += 4) {
								const Intv<P> v0 = lst[k], v1 = lst[k + 1], v2 = lst[k + 2], v3 = lst[k + 3];
								add_aln(v0.L, v0.U, e_score, alen2); add_aln(v1.L, v1.U, e_score, alen2);
								add_aln(v2.L, v2.U, e_score, alen2); add_aln(v3.L, v3.U, e_score, alen2);
							}
							const int rem = curT - 1 - k; /* 0..3 intervals of the list in memory are left */
							if (rem > 0 && n_alns + rem <= (int)sc_acap) {
								Intv<P> v0 = lst[k], v1 = v0, v2 = v0;
								if (rem > 1) v1 = lst[k + 1];
								if (rem > 2) v2 = lst[k + 2];
								add_aln(v0.L, v0.U, e_score, alen2);
								if (rem > 1) add_aln(v1.L, v1.U, e_score, alen2);
								if (rem > 2) add_aln(v2.L, v2.U, e_score, alen2);
								k += rem;
							}
						}
						for (; k < curT && !ovf; k++) {
							P L, U;
							if (k == curT - 1) { L = cL; U = cU; }
							else { const Intv<P> v = lst[k]; L = v.L; U = v.U; }
							add_aln(L, U, e_score, alen2);
						}
					}
				}
			}
		}

		STAMP(15);
		HIST(H_TOP_RELOAD, ld_cnt != 0);
		/* (A top that is still missing - something else than a match went on top of the cached bucket, equal or zero penalties only - is
		 * fetched by the next pop, which waits for it inside its branch.  Rounds 3-5 issued that load here, to have it under way: an ordinary
		 * load that the compiler sees pending on the loop's back edge, so it put an `s_waitcnt vmcnt(0)` in front of the pop of EVERY
		 * iteration - a wait for the iteration's stores, a memory round trip that a wave with its SIMD to itself sits out in full, for a
		 * load that one iteration in a hundred issues: tools/bbprof.py listing, session 9.) */
		/* heap entries stored / fetched: two per-lane accumulators, summed over the wave ONCE per launch (after the loop: one LDS atomic whose
		 * index goes through a vector register the compiler knows nothing about - with an address it can prove wave-uniform it replaces the
		 * atomic by a 64-trip scalar loop).  Round 4 did that atomic in every iteration: 64 lanes on one LDS address were the bank conflicts
		 * of profiles/r4_c3_pmc_sq_final.txt; +0.7 % reads/s without them (profiles/r4_ab_steps.txt session 14). */
		acc_st += st_cnt; acc_ld += ld_cnt;
		STAMP(5);
		if (ovf) finish = true;
		HIST(H_FINISH, finish);
		if (finish) {
			h.fhead = NONE32;
			const SlotDesc &d = R_descs[rd_myslot]; /* (the read may belong to an earlier slot than the one this launch feeds from) */
			const OutBuf out = d.out;
			unsigned long long off = 0;
			bool outovf = false;
			if (!ovf && n_alns > 0) {
				off = atomicAdd(out.count, (unsigned long long)n_alns);
				if (off + (unsigned long long)n_alns > out.cap) outovf = true;
				else for (int t = 0; t < n_alns * ALN_U4; t++) out.alns[off * ALN_U4 + t] = myalns[t];
			}
			out.off[rid] = off;
			out.n[rid] = (ovf || outovf) ? 0u : (uint32_t)n_alns;
			d.b.status[rid] = ovf ? ST_SCRATCH_OVF : (outovf ? ST_OUT_OVF : ST_OK);
			if (d.b.dbg_iters) d.b.dbg_iters[rid] = r_pop + r_vis_s / 2; /* (developer aid: about the loop iterations the read took: pops + exact steps) */
			if (!ovf && !outovf) { /* (a read finishes once in tens of thousands of iterations: five atomics instead of ten registers held across the loop) */
				if (n_alns) atomicAdd(&R_stats[STAT_ALNS], (unsigned long long)n_alns);
				if (r_vis_s) atomicAdd(&R_stats[STAT_VIS_SINGLE], (unsigned long long)r_vis_s);
				if (r_vis_a) atomicAdd(&R_stats[STAT_VIS_ALPHA], (unsigned long long)r_vis_a);
				atomicAdd(&R_stats[STAT_POPS], (unsigned long long)r_pop);
				atomicAdd(&R_stats[STAT_PUSHES], (unsigned long long)r_push);
			}
			publish_done(d.done);
			/* leave every bucket state empty for the next read and give its chunks back */
			const uint32_t xchunks = h.release_excess(xs);
			__hip_atomic_fetch_add((Lds<unsigned int>)&s_need_sum, (h.pused + xchunks + 15u) >> 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_add((Lds<unsigned int>)&s_need_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			__hip_atomic_fetch_add((Lds<unsigned int>)&s_active, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			h.bstate[h.cb] = NONE32;
			if (WIDE && h.far) { for (int k = 0; k < nb; k++) h.bstate[k] = NONE32; h.neW = 0; } /* (buckets beyond the window may be in use) */
			while (h.neW) { const int k = h.best(nb); h.bstate[k] = NONE32; h.unmark(k); }
			active = false;
		}
	}
	if (lane == 0 && n_bkt) atomicAdd(&R_stats[STAT_BKT_SEARCH], (unsigned long long)n_bkt);
	{ uint32_t wv = wave_in_block; asm volatile("" : "+v"(wv));
	  __hip_atomic_fetch_add((Lds<unsigned long long>)&s_cnt[wv], (unsigned long long)acc_st | ((unsigned long long)acc_ld << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
	if (lane == 0) { const unsigned long long sc_ = __hip_atomic_load((Lds<unsigned long long>)&s_cnt[wave_in_block], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicAdd(&R_stats[STAT_ENT_ST], sc_ & 0xFFFFFFFFull); atomicAdd(&R_stats[STAT_ENT_LD], sc_ >> 32); atomicAdd(&R_stats[STAT_REC_LD], (unsigned long long)n_rec); }
	if (parked) atomicAdd(&R_stats[STAT_PARKED], 1ull);
	if (lane == 0) {
		atomicAdd(&R_stats[STAT_N], (unsigned long long)n_iter);            /* loop iterations of busy lanes */
		atomicAdd(&R_stats[STAT_WAVE_ITERS], (unsigned long long)w_iter);   /* loop iterations of waves: the ratio = lanes busy of 64 */
	}
#ifdef BWB_STAMPS
	for (int k = 0; k < 16; k++) if (seg[k]) atomicAdd(&R_stats[STAT_STAMPS + k], seg[k]);
#endif
#ifdef BWB_HIST
	hist[H_PUSH_GAP] = 0; hist[H_PUSH_MIS] = 0; hist[H_PUSH_MATCH] = 0;
	if ((threadIdx.x & 63u) == 0) for (int k = 0; k < H_N; k++) if (hist[k]) atomicAdd(&R_stats[STAT_HIST + k], hist[k]);
	if (hl_gap) atomicAdd(&R_stats[STAT_HIST + H_PUSH_GAP], hl_gap);
	if (hl_mis) atomicAdd(&R_stats[STAT_HIST + H_PUSH_MIS], hl_mis);
	if (hl_match) atomicAdd(&R_stats[STAT_HIST + H_PUSH_MATCH], hl_match);
	if (hl_phg) atomicAdd(&R_stats[STAT_HIST + H_PH_GAP], hl_phg);
	if (hl_phx) atomicAdd(&R_stats[STAT_HIST + H_PH_MIS], hl_phx);
#endif
	if (!parked) mysave[0].x = 0u;
	/* the last wave of the block to leave hands the block's recycle stack to the next slice */
	unsigned int left = 0;
	if ((threadIdx.x & 63u) == 0) left = __hip_atomic_fetch_add((Lds<unsigned int>)&s_left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	if ((threadIdx.x & 63u) == 0 && left == LANE_BLOCK / 64 - 1) {
		const unsigned long long v = ~__hip_atomic_load((Lds<unsigned long long>)&s_blockfree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		uint32_t *bs = R_sc(blocksave) + (size_t)blockIdx.x * 4;
		bs[0] = (uint32_t)v; bs[1] = (uint32_t)(v >> 32); bs[2] = __hip_atomic_load((Lds<unsigned int>)&s_nfree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}

#undef ERUNS_LO
#undef ERUNS_HI
#undef ERUNS2_LO
#undef ERUNS2_HI
#undef KARGS
#undef R_descs
#undef R_wk
#undef R_stats
#undef R_sc
#undef myalns
#undef mysave
#undef e_score
#undef cL
#undef cU
#undef rd_len
#undef rd_max_diff
#undef rd_myslot
#undef rd_best_score
#undef SET_LEN
#undef SET_MAX_DIFF
#undef SET_MYSLOT
#undef SET_BEST_SCORE

/* Rank micro-benchmark, lane layout: one query per lane - the wave gathers the 64 buckets cooperatively (wave_gather, L rows
 * only) and every lane ranks all 15 codes of its own bucket from LDS (block_pops): the access pattern and the ALU work of a rank
 * visit in kl_search / kl_calc_d without anything else.  Same queries and same checksum as k_rank_bench (octet layout). */
template <typename P>
__global__ __launch_bounds__(LANE_BLOCK) void k_rank_bench_lane(DevIndex ix, uint64_t n, uint64_t seed, unsigned long long *checksum) {
	__shared__ P s_base[BWB_BASE_ROWS * 16];
	__shared__ __align__(128) u32x4 s_zero[8];
	__shared__ __align__(128) u32x4 s_stage[LANE_BLOCK / 64][WAVE_LDS_BYTES / 16]; /* rows + exchange array (rows 128-byte aligned: RowRef) */
	if (threadIdx.x < 32) ((Lds<uint32_t>)&s_zero[0])[threadIdx.x] = 0u;
	load_base<P>(s_base, ix);
	const uint64_t nl = (uint64_t)gridDim.x * LANE_BLOCK;
	const int lane = (int)(threadIdx.x & 63u);
	Lds<u32x4> stage = (Lds<u32x4>)&s_stage[threadIdx.x >> 6][0], zero_row = (Lds<u32x4>)&s_zero[0];
	const P last_row = (P)(ix.length - 1);
	unsigned long long acc = 0;
	const uint64_t n_round = (n + nl - 1) / nl * nl; /* whole waves iterate together */
	for (uint64_t q = (uint64_t)blockIdx.x * LANE_BLOCK + threadIdx.x; q < n_round; q += nl) {
		uint64_t x = q * 0x9E3779B97F4A7C15ull + seed;
		x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
		const P pos = (P)(x % (ix.length - 1));
		PairInfo<P> pi;
		pair_setup<P>(last_row, q < n, pos, pos, lane, pi);
		wave_gather<P>(ix.buckets, pi, 0, stage, lane);
		uint32_t rel[16];
		const bool own = pi.blkL != NONE32;
		const RowRef row = own ? row_ref(stage, (uint32_t)lane) : (uint32_t)(uintptr_t)zero_row;
		{
			SideBits bl;
			side_read(row, pi.offL, bl);
			sub_pops16(bl.planes, bl.n, rel);
			const uint32_t md[4] = { bl.mid.x, bl.mid.y, bl.mid.z, bl.mid.w };
#pragma unroll
			for (int s = 0; s < 4; s++) {
				const u32x4 c4 = *row_slice(row, (uint32_t)s << 4);
				rel[2 * s] += c4.x + (md[s] & 255u); rel[2 * s + 1] += c4.y + ((md[s] >> 8) & 255u);
				rel[8 + 2 * s] += c4.z + ((md[s] >> 16) & 255u); rel[8 + 2 * s + 1] += c4.w + (md[s] >> 24);
			}
		}
		if (q < n) {
			const P *brow = s_base + pi.rowL * 16;
#pragma unroll
			for (int j = 1; j < 16; j++) acc += (unsigned long long)(P)(brow[j] + (P)rel[j]) * ((j & 1) ? 3ull : 1ull);
		}
	}
	for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
	if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(checksum, acc);
}
