/*
 * dropin_main.c - TEST DRIVER for integration/align_gpu.c, linked with the reference's own objects (oracle/Makefile ->
 * oracle/_ref/bwbble_dropin).  It does what align_reads (mg-aligner/align.c:40-87) does with the GPU branch of INTEGRATION.md
 * section 3 taken: the reference's load_bwt and fastq2reads build the reference's bwt_t / reads_t, the binding aligns them.
 *
 *     bwbble_dropin <seq_fasta> <reads_fastq> <output_aln> [-n max_diff] [-o max_gapo] [-e max_gape] [-l seed] [-k seed_diff] [-S] [-P]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwt.h"
#include "io.h"
#include "align.h"
#include "inexact_match.h"

int align_reads_inexact_gpu(bwt_t *BWT, reads_t *reads, sa_intv_list_t *precalc, aln_params_t *params, char *alnFname);

int main(int argc, char **argv) {
	if (argc < 4) { printf("usage: bwbble_dropin <seq_fasta> <reads_fastq> <output_aln> [flags]\n"); return 1; }
	aln_params_t *params = (aln_params_t *)calloc(1, sizeof(aln_params_t));
	set_default_aln_params(params);
	for (int i = 4; i < argc; i++) {
		if (!strcmp(argv[i], "-S")) params->is_multiref = 0;
		else if (!strcmp(argv[i], "-P")) params->use_precalc = 1;
		else if (i + 1 < argc && !strcmp(argv[i], "-n")) params->max_diff = atoi(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-o")) params->max_gapo = atoi(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-e")) params->max_gape = atoi(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-l")) params->seed_length = atoi(argv[++i]);
		else if (i + 1 < argc && !strcmp(argv[i], "-k")) params->max_diff_seed = atoi(argv[++i]);
		else { printf("unknown flag %s\n", argv[i]); return 1; }
	}
	char *bwtFname = (char *)malloc(strlen(argv[1]) + 5);
	sprintf(bwtFname, "%s.bwt", argv[1]);
	remove(argv[3]);
	bwt_t *BWT = load_bwt(bwtFname, 0);
	reads_t *reads = fastq2reads(argv[2]);
	align_reads_inexact_gpu(BWT, reads, NULL, params, argv[3]);
	free_bwt(BWT);
	free(bwtFname);
	free(params);
	return 0;
}
