/*
 * align_gpu.c - the reference-side binding of libbwbble_hip.so: drop this file into viq854/bwbble's mg-aligner/ next to
 * inexact_match.c.  It is written against the REFERENCE's own headers (bwt.h, io.h, align.h, inexact_match.h) and gives
 *
 *     int align_reads_inexact_gpu(bwt_t *BWT, reads_t *reads, sa_intv_list_t *precalc, aln_params_t *params, char *alnFname);
 *
 * the signature and contract of align_reads_inexact[_parallel] (mg-aligner/inexact_match.h:39-40, inexact_match.c:25-168):
 * one .aln record per read appended in input order through the reference's own alns2alnf_bin (align.c:345-382), returns 0,
 * errors printf + exit(1).  INTEGRATION.md shows the three lines that select it in align_reads (align.c:72-76).
 *
 * tests/test_dropin_binding.py compiles this file with -I<reference>/mg-aligner, links it with the reference's objects and
 * runs the result on the GPU against the reference's golden .aln files.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwt.h"
#include "io.h"
#include "align.h"
#include "inexact_match.h"
#include "bwbble_hip.h"

/* bwb_params IS aln_params_t (align.h:48-79): same fields, same order, same size - the pointer is passed through */
#define SAME_FIELD(f) _Static_assert(offsetof(bwb_params, f) == offsetof(aln_params_t, f), "bwb_params differs from aln_params_t at " #f)
_Static_assert(sizeof(bwb_params) == sizeof(aln_params_t), "bwb_params and aln_params_t differ in size");
SAME_FIELD(max_diff); SAME_FIELD(max_gapo); SAME_FIELD(max_gape); SAME_FIELD(max_entries); SAME_FIELD(mm_score); SAME_FIELD(gapo_score);
SAME_FIELD(gape_score); SAME_FIELD(seed_length); SAME_FIELD(max_diff_seed); SAME_FIELD(max_best); SAME_FIELD(no_indel_length);
SAME_FIELD(matched_Ncontig); SAME_FIELD(use_precalc); SAME_FIELD(is_multiref); SAME_FIELD(n_threads);
_Static_assert(sizeof(bwtint_t) == sizeof(uint64_t), "bwtint_t must be 64-bit (bwt.h)");

#define GPU_BATCH (1u << 21)  /* reads per GPU batch; the reference's READ_BATCH_SIZE (align.h:14) only paces its output */
#define GPU_SLOTS BWB_MAX_SLOTS

static void gpu_die(const char *what) {
	printf("align_reads_inexact_gpu: %s: %s\n", what, bwb_hip_last_error());
	exit(1);
}

/* one hit of the library -> one aln_t of the read, through the reference's own add_alignment (align.c:271-298): the edit
 * path is aln_length bytes of STATE_M except for the gap runs (include/bwbble_hip.h: start | len << 8 | is_deletion << 15) */
static void add_bwb_aln(alns_t *alns, const bwb_aln *a, const aln_params_t *params) {
	aln_entry_t e;
	memset(&e, 0, sizeof(e));
	e.num_mm = a->num_mm; e.num_gapo = a->num_gapo; e.num_gape = a->num_gape; e.aln_length = a->aln_length;
	for (int k = 0; k < BWB_MAX_GAP_RUNS; k++) {
		const unsigned run = a->gap_run[k];
		if (run == 0xFFFFu) continue;
		memset(e.aln_path + (run & 0xFF), (run >> 15) ? STATE_D : STATE_I, (run >> 8) & 0x7F);
	}
	add_alignment(&e, a->L, a->U, a->score, alns, params);
}

/* the hits of one finished batch -> the reads' alns_t -> .aln, in input order (inexact_match.c:154-162) */
static void write_batch(bwb_hip_ctx *ctx, int slot, reads_t *reads, unsigned first, const aln_params_t *params, FILE *alnFile) {
	bwb_result res;
	if (bwb_hip_slot_result(ctx, slot, &res)) gpu_die("slot_result");
	for (uint32_t i = 0; i < res.n_reads; i++) {
		read_t *read = &reads->reads[first + i];
		read->alns = init_alignments();
		for (uint64_t k = res.aln_off[i]; k < res.aln_off[i + 1]; k++) add_bwb_aln(read->alns, &res.alns[k], params);
		alns2alnf_bin(read->alns, alnFile);
		free_alignments(read->alns);
		read->alns = NULL;
	}
	printf("Processed %u reads.\n", first + res.n_reads);
}

int align_reads_inexact_gpu(bwt_t *BWT, reads_t *reads, sa_intv_list_t *precalc_sa_intervals_table, aln_params_t *params, char *alnFname) {
	(void)precalc_sa_intervals_table; /* -P: a table entry is exact_match() of the read's last 12 bases (align.c:212-216); the kernel computes it */
	printf("BWBBLE Inexact Alignment (MI355X)...\n");
	FILE *alnFile = (FILE *)fopen(alnFname, "a+b");
	if (alnFile == NULL) {
		printf("align_reads_inexact: Cannot open ALN file: %s!\n", alnFname);
		perror(alnFname);
		exit(1);
	}
	bwb_hip_ctx *ctx = NULL;
	const uint64_t hdr[5] = { BWT->length, BWT->num_words, BWT->num_sa, BWT->num_occ, BWT->sa0_index };
	if (bwb_hip_abi_version() != BWB_HIP_ABI_VERSION) { printf("align_reads_inexact_gpu: libbwbble_hip.so implements C-ABI version %d, this binding was compiled against %d\n", bwb_hip_abi_version(), BWB_HIP_ABI_VERSION); exit(1); }
	if (bwb_hip_ctx_create(0, hdr, BWT->C, BWT->bwt, BWT->O, &ctx)) gpu_die("ctx_create");

	/* read_t keeps one malloc per read (io.h:151-185): pack read->seq codes into one [batch][max_len] array per batch */
	const unsigned stride = reads->max_len ? reads->max_len : 1;
	uint8_t *seq = (uint8_t *)malloc((size_t)GPU_BATCH * stride);
	uint16_t *len = (uint16_t *)malloc(sizeof(uint16_t) * GPU_BATCH);
	if (!seq || !len) { printf("align_reads_inexact_gpu: out of memory\n"); exit(1); }
	unsigned first_of[GPU_SLOTS];
	unsigned submitted = 0, j = 0; /* reads handed to the GPU; batches submitted */
	while (submitted < reads->count) {
		const unsigned batch = reads->count - submitted > GPU_BATCH ? GPU_BATCH : reads->count - submitted;
		for (unsigned i = 0; i < batch; i++) {
			const read_t *r = &reads->reads[submitted + i];
			len[i] = (uint16_t)(r->len > 0xFFFF ? 0xFFFF : r->len);
			memcpy(seq + (size_t)i * stride, r->seq, (size_t)r->len);
		}
		const int slot = (int)(j % GPU_SLOTS);
		/* slot_upload copies the reads (seq/len are free again); slot_submit queues calculate_d + one search slice and returns */
		/* a read at the head of the batch that is not longer than the seed sees the D_seed of the last longer read before it
		 * (the serial reference's one D_seed buffer, inexact_match.c:35,62-65): hand that read over */
		const char *carry = NULL;
		unsigned carry_len = 0;
		for (unsigned q = submitted; q-- > 0;) { /* (no bound: the serial reference has none; stops at the first longer read) */
			const read_t *r = &reads->reads[q];
			int ok = params->seed_length && r->len > params->seed_length && r->len <= 255;
			if (ok && params->use_precalc && (r->len < PRECALC_INTERVAL_LENGTH || read2index(r->rc, r->len) < 0)) ok = 0; /* dropped before calculate_d (:50-57) */
			if (ok) { carry = r->seq; carry_len = (unsigned)r->len; break; }
		}
		if (bwb_hip_slot_upload(ctx, slot, (const bwb_params *)params, seq, len, batch, stride, (const uint8_t *)carry, carry_len) || bwb_hip_slot_submit(ctx, slot)) gpu_die("submit");
		first_of[slot] = submitted;
		submitted += batch;
		j++;
		if (j >= GPU_SLOTS) write_batch(ctx, (int)((j - GPU_SLOTS) % GPU_SLOTS), reads, first_of[(j - GPU_SLOTS) % GPU_SLOTS], params, alnFile);
	}
	for (unsigned k = j >= GPU_SLOTS - 1 ? j - (GPU_SLOTS - 1) : 0; k < j; k++) write_batch(ctx, (int)(k % GPU_SLOTS), reads, first_of[k % GPU_SLOTS], params, alnFile);
	free(seq); free(len);
	bwb_hip_ctx_destroy(ctx);
	fclose(alnFile);
	return 0;
}
