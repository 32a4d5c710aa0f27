/*
 * sam.c - `bwbble aln2sam`: .aln + FASTQ -> SAM, same text as the reference (mg-aligner/align.c:494-556 alns2sam,
 * :562-652 print_aln2sam, :738-812 mapq / eval_aln).
 *
 * The only index work here is SA(aln.L) for the reported hit of every mapped read (align.c:786), an invPsi
 * walk of up to 31 dependent rank queries (bwt.c:311-329).  It runs on the GPU for all reads at once
 * (bwb_hip_locate); MAPQ (the only floating point in the tool), CIGAR and text stay on the host.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "bwb_host.h"

#define SAM_FSU 4
#define SAM_FSR 16

static int mapq(int top1, int top2, int num_mm, int max_mm) { /* align.c:738-746 */
	if (top1 == 0) return 23;
	if (top1 > 1) return 0;
	if (num_mm == max_mm) return 25;
	if (top2 == 0) return 37;
	int n = top2 >= 255 ? 255 : top2;
	int q = (int)(4.343 * log(n) + 0.5);
	return 23 < q ? 0 : 23 - q;
}

/* SA lookups of a contiguous share of the rows on one GPU (the index is replicated, like in align) */
typedef struct { int device; const bwt_t *BWT; const uint64_t *rows; uint64_t *pos; size_t n; uint64_t steps; double kernel_ms; } locate_job_t;
static void *locate_worker(void *arg) {
	locate_job_t *j = (locate_job_t *)arg;
	bwb_hip_ctx *ctx = NULL;
	const bwtint_t hdr[5] = { j->BWT->length, j->BWT->num_words, j->BWT->num_sa, j->BWT->num_occ, j->BWT->sa0_index };
	if (bwb_hip_ctx_create(j->device, hdr, j->BWT->C, j->BWT->bwt, j->BWT->O, &ctx) || bwb_hip_set_sa(ctx, j->BWT->SA, j->BWT->num_sa) ||
	    bwb_hip_locate(ctx, j->rows, j->n, j->pos))
		bwb_die("alns2sam: GPU %d: %s", j->device, bwb_hip_last_error());
	bwb_hip_locate_stats(ctx, NULL, &j->steps, &j->kernel_ms);
	bwb_hip_ctx_destroy(ctx);
	return NULL;
}

void alns2sam(char *fastaFname, char *readsFname, char *alnsFname, char *samFname, int is_multiref, int max_diff, int n_gpus) {
	(void)is_multiref;
	printf("**** BWBBLE Alignment Evaluation/SAM File Generation ****\n");
	size_t Ln = strlen(fastaFname) + 8;
	char *bwtFname = (char *)malloc(Ln), *annFname = (char *)malloc(Ln);
	snprintf(bwtFname, Ln, "%s.bwt", fastaFname);
	snprintf(annFname, Ln, "%s.ann", fastaFname);
	bwt_t *BWT = load_bwt(bwtFname, 1);
	fasta_annotations_t *ann = annf2ann(annFname);
	alns_batch_t *alns = alnsf2alns_bin(alnsFname);
	reads_t *reads = fastq2reads(readsFname);
	FILE *sam = fopen(samFname, "w");
	if (!sam) { perror(samFname); bwb_die("alns2sam: Cannot open SAM file: %s!", samFname); }
	for (int i = 0; i < ann->num_seq; i++)                                           /* align.c:522-525 */
		fprintf(sam, "@SQ\tSN:%s\tLN:%d\n", ann->seq_anns[i].name, (int)(ann->seq_anns[i].end_index - ann->seq_anns[i].start_index + 1));
	fprintf(sam, "@PG\tID:bwbble\tPN:bwbble\tVN:0.1-r01\n");

	const size_t n = reads->count < alns->n_reads ? reads->count : alns->n_reads;    /* align.c:535-537 */
	/* SA(aln.L) of the first entry of every mapped read, on the GPU */
	uint64_t *rows = (uint64_t *)malloc((n ? n : 1) * 8), *pos = (uint64_t *)malloc((n ? n : 1) * 8);
	size_t *which = (size_t *)malloc((n ? n : 1) * sizeof(size_t));
	size_t nm = 0;
	for (size_t r = 0; r < n; r++)
		if (alns->aln_off[r + 1] > alns->aln_off[r]) { rows[nm] = alns->alns[alns->aln_off[r]].L; which[nm] = r; nm++; }
	if (nm) {
		const int ndev = bwb_hip_device_count();
		if (ndev < 1) bwb_die("alns2sam: no HIP device found (SA lookups run on the GPU)");
		if (n_gpus < 1) n_gpus = 1;
		if (n_gpus > ndev) bwb_die("alns2sam: -g %d asked for, %d HIP device(s) available", n_gpus, ndev);
		if ((size_t)n_gpus > nm) n_gpus = (int)nm;
		locate_job_t jobs[64];
		pthread_t th[64];
		if (n_gpus > 64) n_gpus = 64;
		for (int g = 0; g < n_gpus; g++) { /* contiguous shares, like the read chunks of align (inexact_match.c:115-116) */
			const size_t lo = (size_t)g * nm / (size_t)n_gpus, hi = (size_t)(g + 1) * nm / (size_t)n_gpus;
			jobs[g] = (locate_job_t){ .device = g, .BWT = BWT, .rows = rows + lo, .pos = pos + lo, .n = hi - lo };
			if (pthread_create(&th[g], NULL, locate_worker, &jobs[g])) bwb_die("alns2sam: cannot start a host thread");
		}
		for (int g = 0; g < n_gpus; g++) pthread_join(th[g], NULL);
		uint64_t steps = 0; double kms = 0;
		for (int g = 0; g < n_gpus; g++) { steps += jobs[g].steps; if (jobs[g].kernel_ms > kms) kms = jobs[g].kernel_ms; }
		printf("SA lookups on the GPU: rows %zu  rank-block visits %llu  kernel %.3f ms  (%.2f G visits/s)\n", nm, (unsigned long long)steps, kms, kms > 0 ? steps / kms / 1e6 : 0.0);
	}
	uint64_t *ref_pos = (uint64_t *)calloc(n ? n : 1, 8);
	for (size_t k = 0; k < nm; k++) ref_pos[which[k]] = pos[k];

	int ann_sorted = 1; /* records as fasta2ref writes them: increasing, disjoint */
	for (int i = 1; i < ann->num_seq; i++) if (ann->seq_anns[i].start_index <= ann->seq_anns[i - 1].end_index) ann_sorted = 0;
	/* The text: blocks of reads are formatted into memory by all cores and written in order (round 5: one thread's fprintf calls were
	 * the wall time of aln2sam on a 10 M-read file, not the SA lookups). */
	const size_t BLK = 1u << 14;
	const size_t nblk = (n + BLK - 1) / BLK;
	const size_t WAVE = 64; /* blocks formatted before the writer writes them */
	char **bufs = (char **)calloc(WAVE, sizeof(char *));
	size_t *lens = (size_t *)calloc(WAVE, sizeof(size_t));
	for (size_t b0 = 0; b0 < nblk; b0 += WAVE) {
		const size_t nb_ = nblk - b0 < WAVE ? nblk - b0 : WAVE;
#pragma omp parallel for schedule(dynamic, 1) num_threads(bwb_host_team())
		for (long bi = 0; bi < (long)nb_; bi++) {
			const size_t r0 = (b0 + (size_t)bi) * BLK, r1 = r0 + BLK < n ? r0 + BLK : n;
			size_t cap = 0;
			for (size_t r = r0; r < r1; r++) cap += (size_t)reads->name_len[r] + 2 * (size_t)reads->len[r] + MAX_SEQ_NAME_LEN + 160;
			char *o = (char *)malloc(cap ? cap : 1), *o0 = o;
			unsigned char path[272];
			for (size_t r = r0; r < r1; r++) {
				const bwb_aln *e = alns->alns + alns->aln_off[r];
				const uint64_t ne = alns->aln_off[r + 1] - alns->aln_off[r];
				const int len = reads->len[r];
				const uint8_t *seq = reads->seq + (size_t)r * reads->stride;
				const char *name = reads->raw + reads->name_off[r];
				const char *qual = reads->raw + reads->qual_off[r];
				memcpy(o, name, reads->name_len[r]); o += reads->name_len[r];
				if (ne == 0) { /* unmapped, align.c:629-651 (aln_strand is 0 for a read that was never evaluated) */
					o += sprintf(o, "\t%d\t*\t0\t0\t*\t*\t0\t0\t", SAM_FSU);
					for (int i = 0; i < len; i++) o[i] = "AGCTN"[seq[i]];
					o += len; *o++ = '\t';
					memcpy(o, qual, (size_t)len); o += len; *o++ = '\n';
					continue;
				}
				/* eval_aln, align.c:760-812 */
				int top1 = 0, top2 = 0;
				const int best_score = e[0].score;
				for (uint64_t i = 0; i < ne; i++) {
					if (e[i].score > best_score) top2 += (int)(e[i].U - e[i].L + 1);
					else top1 += (int)(e[i].U - e[i].L + 1);
				}
				int alen = aln_path_bytes(&e[0], path);
				int ref_len = alen;                                                          /* get_aln_length :748-757 */
				for (int i = 0; i < alen; i++) if (path[i] == 1) ref_len--;
				const uint64_t rp = ref_pos[r];
				int strand;
				uint64_t aln_pos;
				if (rp > (BWT->length - 1) / 2) { strand = 0; aln_pos = ((BWT->length - 1) - rp - 1) - (uint64_t)ref_len + 1; }
				else { strand = 1; aln_pos = rp; }
				const int mq = mapq(top1, top2, e[0].num_mm, max_diff);
				/* the record that contains aln_pos (align.c:796-801 scans linearly; a multi-genome has a record per bubble - 1.3 M at
				 * GRCh37 scale - and the records are disjoint and in text order, so a binary search finds the same one) */
				int seqid = -1;
				if (ann_sorted) {
					int lo = 0, hi = ann->num_seq - 1;
					while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ann->seq_anns[mid].start_index <= aln_pos) lo = mid; else hi = mid - 1; }
					if (ann->num_seq > 0 && aln_pos >= ann->seq_anns[lo].start_index && aln_pos <= ann->seq_anns[lo].end_index) seqid = lo;
				} else
					for (int i = 0; i < ann->num_seq; i++)
						if (aln_pos >= ann->seq_anns[i].start_index && aln_pos <= ann->seq_anns[i].end_index) { seqid = i; break; }
				if (seqid < 0) bwb_die("alns2sam: read %zu maps outside every annotated sequence", r); /* the reference indexes seq_anns[-1] here */
				o += sprintf(o, "\t%d\t%s\t%d\t%d\t", strand ? SAM_FSR : 0, ann->seq_anns[seqid].name, (int)(aln_pos - ann->seq_anns[seqid].start_index + 1), mq);
				if (strand) for (int i = 0; i < alen >> 1; i++) { unsigned char t = path[alen - 1 - i]; path[alen - 1 - i] = path[i]; path[i] = t; }
				/* CIGAR: runs of the path walked from its end to its start (align.c:588-609) */
				int i = alen - 1;
				while (i >= 0) {
					int j = i;
					while (j >= 0 && path[j] == path[i]) j--;
					o += sprintf(o, "%d%c", i - j, "MID"[path[i]]);
					i = j;
				}
				memcpy(o, "\t*\t0\t0\t", 7); o += 7;
				if (strand) for (int k = 0; k < len; k++) { const int c = seq[len - 1 - k]; o[k] = "AGCTN"[c > 3 ? 4 : 3 - c]; } /* read->rc */
				else for (int k = 0; k < len; k++) o[k] = "AGCTN"[seq[k]];
				o += len; *o++ = '\t';
				if (strand) for (int k = 0; k < len; k++) o[k] = qual[len - 1 - k];
				else memcpy(o, qual, (size_t)len);
				o += len; *o++ = '\n';
			}
			bufs[bi] = o0; lens[bi] = (size_t)(o - o0);
		}
		for (size_t bi = 0; bi < nb_; bi++) {
			if (lens[bi] && fwrite(bufs[bi], 1, lens[bi], sam) != lens[bi]) bwb_die("alns2sam: Cannot write to the SAM file: %s!", samFname);
			free(bufs[bi]); bufs[bi] = NULL;
		}
	}
	free(bufs); free(lens);
	printf("Processed %zu reads.\n", n);
	free(rows); free(pos); free(which); free(ref_pos);
	free(bwtFname); free(annFname);
	free_bwt(BWT); free_reads(reads); free_alns_batch(alns); free_ann(ann);
	fclose(sam);
}
