/*
 * precalc.c - the `.pre` table of `align -P` (mg-aligner/align.c:174-238): for each of the 4^12 = 16 777 216 12-mers the list of
 * SA intervals of its exact matches (exact_match, exact_match.c), stored as `int size` + size x {u64 L, u64 U} per list
 * (store_sa_interval_list, align.c:141-150), lists in the order of next_read (align.c:187-198): the 12-mer is a base-4 counter
 * over A0 G1 C2 T3 whose LAST position runs fastest.
 *
 * The GPU search does not need the table: with -P it runs the 12 exact steps of a read itself (bwb_lane.h), which gives the
 * list the table would hold (it is a pure function of the index) without 16.7 M lists in HBM.  But the reference WRITES
 * <fasta>.pre on the first `align -P` (align.c:59-65) and other runs of the reference READ it, so this build writes the same
 * bytes when the file is missing: every 12-mer goes through the library as a 12-base read with `-n 0` - the root entry has no
 * difference left, so the search is exactly exact_match_bounded from the whole index (inexact_match.c:345-347), one hit per
 * interval of the final list, in list order (:366-370); a 12-mer that cannot match is dropped by its D bound and gets size 0.
 * An existing .pre is left alone; its CONTENT is not used (nothing in it could change a result), but its shape is checked the way
 * load_precalc_sa_intervals (align.c:226-238) would walk it: 16 777 216 lists of `int size` + size x 16 bytes must end exactly at the
 * end of the file - a truncated table (a crashed earlier run) would make the reference fail, so it is reported.
 * The table is written to <fasta>.pre.tmp and renamed when complete; the context for it is created on the first device of
 * BWB_DEVICE_MAP (default 0) with a small heap pool (12-base reads with -n 0 need next to none).
 */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "bwb_host.h"

#define PRECALC_INTERVAL_LENGTH 12  /* align.h:31 */
#define NUM_PRECALC 16777216u       /* align.h:30 */

/* walks an existing table like load_precalc_sa_intervals (align.c:226-238) without keeping it: 0 = well-formed */
int check_precalc_file(const char *preFname) {
	FILE *f = fopen(preFname, "rb");
	if (!f) return -1;
	static unsigned char buf[1 << 20];
	uint64_t lists = 0, skip = 0; /* bytes of interval records still to pass over */
	unsigned char szb[4]; int have = 0;
	size_t n;
	int bad = 0;
	while (!bad && (n = fread(buf, 1, sizeof(buf), f)) > 0) {
		size_t i = 0;
		while (i < n) {
			if (skip) { const uint64_t k = skip < n - i ? skip : n - i; skip -= k; i += (size_t)k; continue; }
			szb[have++] = buf[i++];
			if (have == 4) {
				int32_t size; memcpy(&size, szb, 4); have = 0;
				if (size < 0 || lists == NUM_PRECALC) { bad = 1; break; }
				lists++; skip = (uint64_t)size * 16u;
			}
		}
	}
	fclose(f);
	return (bad || have || skip || lists != NUM_PRECALC) ? 1 : 0;
}

void precalc_sa_intervals(bwt_t *BWT, const aln_params_t *params, const char *preFname) { /* align.c:200-224 */
	printf("Pre-calculating SA intervals...\n");
	const size_t TL = strlen(preFname) + 8;
	char *tmpFname = (char *)malloc(TL);
	snprintf(tmpFname, TL, "%s.tmp", preFname); /* a run that dies half-way leaves no <fasta>.pre behind for later runs to trust */
	FILE *preFile = fopen(tmpFname, "wb");
	if (!preFile) { fprintf(stderr, "precalc_sa_intervals: Cannot open PRE file %s!\n", tmpFname); exit(1); }
	bwb_hip_ctx *ctx = NULL;
	const bwtint_t hdr[5] = { BWT->length, BWT->num_words, BWT->num_sa, BWT->num_occ, BWT->sa0_index };
	int device = 0;
	if (getenv("BWB_DEVICE_MAP")) device = atoi(getenv("BWB_DEVICE_MAP")); /* (its first entry: the device worker 0 of the alignment will use) */
	const int pool_was_set = getenv("BWB_POOL_GB") != NULL;
	if (!pool_was_set) setenv("BWB_POOL_GB", "2", 1); /* (the library sizes its heap pool with the first batch; this context only: removed again below) */
	if (bwb_hip_ctx_create(device, hdr, BWT->C, BWT->bwt, BWT->O, &ctx)) bwb_die("precalc_sa_intervals: %s", bwb_hip_last_error());
	aln_params_t p;
	bwb_default_params(&p);
	p.max_diff = 0; p.is_multiref = params->is_multiref; /* exact_match honours -S (exact_match.c:26-63) */
	const uint32_t CH = 1u << 21, L = PRECALC_INTERVAL_LENGTH;
	uint8_t *seq = (uint8_t *)malloc((size_t)CH * L);
	uint16_t *len = (uint16_t *)malloc((size_t)CH * 2);
	if (!seq || !len) bwb_die("precalc_sa_intervals: out of memory");
	for (uint32_t i = 0; i < CH; i++) len[i] = (uint16_t)L;
	for (uint32_t base = 0; base < NUM_PRECALC; base += CH) {
		/* the library searches read->rc (the reverse complement of what it is given, io.c:502-504): hand it the reverse
		 * complement of the 12-mer, so that what is searched is the 12-mer itself, like exact_match(read->seq) */
		for (uint32_t i = 0; i < CH; i++) {
			const uint32_t w = base + i;
			for (uint32_t k = 0; k < L; k++) seq[(size_t)i * L + k] = (uint8_t)(3u - ((w >> (2 * k)) & 3u)); /* seq[k] = 3 - w[L-1-k]; w[j] = digit L-1-j */
		}
		bwb_result r;
		if (bwb_hip_align_batch(ctx, &p, seq, len, CH, L, &r)) bwb_die("precalc_sa_intervals: %s", bwb_hip_last_error());
		for (uint32_t i = 0; i < CH; i++) {
			const int size = (int)(r.aln_off[i + 1] - r.aln_off[i]);
			fwrite(&size, sizeof(int), 1, preFile);
			for (uint64_t a = r.aln_off[i]; a < r.aln_off[i + 1]; a++) { fwrite(&r.alns[a].L, 8, 1, preFile); fwrite(&r.alns[a].U, 8, 1, preFile); }
		}
	}
	free(seq); free(len);
	bwb_hip_ctx_destroy(ctx);
	if (!pool_was_set) unsetenv("BWB_POOL_GB");
	if (fclose(preFile)) bwb_die("precalc_sa_intervals: cannot write %s", tmpFname);
	if (rename(tmpFname, preFname)) bwb_die("precalc_sa_intervals: cannot rename %s to %s", tmpFname, preFname);
	free(tmpFname);
}
