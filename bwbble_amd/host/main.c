/* main.c - `bwbble` command line, same commands and flags as the reference (mg-aligner/main.c:38-160),
 * plus -g <n_gpus> on align / aln2sam. */
#define _GNU_SOURCE
#include <getopt.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include "bwb_host.h"

static int usage(void) {
	printf("Usage:   bwbble command [options] \n");
	printf("Command: index    index sequences in the FASTA format\n");
	printf("         align    exact or inexact read alignment (MI355X)\n");
	printf("         fasta2ref    constructs a single linear reference from the input file \n");
	printf("         aln2sam  convert alignment results to SAM file format for single-end mapping\n\n");
	return 1;
}
static int align_usage(void) {
	printf("Usage: bwbble align [options] <seq_fasta> <reads_fastq> <output_aln> \n");
	printf("Options: M    mismatch penalty (default: 3)\n         O    gap open penalty (default: 11) \n         E    gap extend penalty (default: 4) \n");
	printf("         n    maximum number of differences in the alignment (gaps and mismatches) (default: 0)\n");
	printf("         l    length of the seed (seed := first seed_length chars of the read) (default: 32)\n");
	printf("         k    maximum number of differences in the seed (default: 2)\n         o    maximum number of gap opens (default: 1)\n");
	printf("         e    maximum number of gap extends (default: 6) \n         t    accepted for compatibility, ignored (the GPU path has no host threads knob)\n");
	printf("         g    number of GPUs to use (default: 1; more than are present is an error)\n");
	printf("         S    align with a single-genome reference\n         P    use pre-calculated partial alignment results (the GPU computes them per read; <fasta>.pre is written like the reference does when it is missing, and only checked for completeness when present)\n\n");
	return 1;
}

int main(int argc, char *argv[]) {
	if (argc < 2) return usage();
	if (strcmp(argv[1], "index") == 0) {
		if (argc < 3) { printf("Usage: bwbble index [options] <seq_fasta> \n"); exit(1); }
		char *esa = NULL;
		int c;
		while ((c = getopt(argc - 1, argv + 1, "e:")) >= 0) {
			if (c == 'e') esa = optarg; else return 1;
		}
		index_bwt(argv[optind + 1], esa);
	} else if (strcmp(argv[1], "align") == 0) {
		if (argc < 5) { align_usage(); exit(1); }
		aln_params_t params;
		set_default_aln_params(&params);
		int c, n_gpus = 1;
		while ((c = getopt(argc - 1, argv + 1, "M:O:E:n:k:o:e:l:m:t:g:SP")) >= 0) {
			switch (c) {
			case 'M': params.mm_score = atoi(optarg); break;
			case 'O': params.gapo_score = atoi(optarg); break;
			case 'E': params.gape_score = atoi(optarg); break;
			case 'n': params.max_diff = atoi(optarg); break;
			case 'k': params.max_diff_seed = atoi(optarg); break;
			case 'o': params.max_gapo = atoi(optarg); break;
			case 'e': params.max_gape = atoi(optarg); break;
			case 'l': params.seed_length = atoi(optarg); break;
			case 'm': params.max_entries = atoi(optarg); break;
			case 't': params.n_threads = atoi(optarg); break;
			case 'g': n_gpus = atoi(optarg); break;
			case 'S': params.is_multiref = 0; break;
			case 'P': params.use_precalc = 1; break;
			case '?': align_usage(); return 1;
			default: return 1;
			}
		}
		if (argc - 1 - optind < 3) { align_usage(); exit(1); }
		align_reads(argv[optind + 1], argv[optind + 2], argv[optind + 3], &params, n_gpus);
	} else if (strcmp(argv[1], "fasta2ref") == 0) {
		if (argc < 3) { printf("Usage: bwbble fasta2ref <seq_fasta> \n"); exit(1); }
		size_t L = strlen(argv[2]) + 8;
		char *refFname = (char *)malloc(L), *annFname = (char *)malloc(L);
		snprintf(refFname, L, "%s.ref", argv[2]);
		snprintf(annFname, L, "%s.ann", argv[2]);
		unsigned char *seq;
		bwtint_t seqLen;
		fasta2ref(argv[2], refFname, annFname, &seq, &seqLen);
		free(seq); free(refFname); free(annFname);
	} else if (strcmp(argv[1], "aln2sam") == 0) {
		if (argc < 6) { printf("Usage: bwbble aln2sam [-S, -n, -g] <seq_fasta> <reads_fastq> <alns_aln> <out_sam> \n"); exit(1); }
		int is_multiref = 1, max_diff = 6, n_gpus = 1, c;
		while ((c = getopt(argc - 1, argv + 1, "n:g:S")) >= 0) {
			switch (c) {
			case 'S': is_multiref = 0; break;
			case 'n': max_diff = atoi(optarg); break;
			case 'g': n_gpus = atoi(optarg); break;
			case '?': printf("Unknown option \n"); break;
			default: return 1;
			}
		}
		alns2sam(argv[optind + 1], argv[optind + 2], argv[optind + 3], argv[optind + 4], is_multiref, max_diff, n_gpus);
	} else if (strcmp(argv[1], "dumpreads") == 0) {
		/* developer command (CPU only, used by the tests): what fastq2reads made of a FASTQ - per read "name<TAB>codes<TAB>quality" */
		if (argc < 4) { printf("Usage: bwbble dumpreads <reads_fastq> <out_tsv> [chunk_reads] \n"); exit(1); }
		if (argc >= 5) { /* the streaming reader of `align` (fq_next_chunk), chunk by chunk: "codes" per read */
			fq_stream *fs = fq_open(argv[2]);
			FILE *g = fopen(argv[3], "w");
			if (!g) { perror(argv[3]); return 1; }
			fq_chunk_t ch;
			while (fq_next_chunk(fs, (uint32_t)atoi(argv[4]), &ch)) {
				for (uint32_t r = 0; r < ch.n; r++) {
					for (uint32_t i = 0; i < ch.stride; i++) if (i < ch.len[r]) fputc('0' + ch.seq[(size_t)r * ch.stride + i], g); else if (ch.seq[(size_t)r * ch.stride + i] != 4) fputc('!', g);
					fputc('\n', g);
				}
				free(ch.seq); free(ch.len);
			}
			fclose(g);
			fq_close(fs);
			return 0;
		}
		reads_t *reads = fastq2reads(argv[2]);
		FILE *f = fopen(argv[3], "w");
		if (!f) { perror(argv[3]); return 1; }
		for (unsigned r = 0; r < reads->count; r++) {
			fprintf(f, "%.*s\t", (int)reads->name_len[r], reads->raw + reads->name_off[r]);
			for (int i = 0; i < reads->len[r]; i++) fputc('0' + reads->seq[(size_t)r * reads->stride + i], f);
			fprintf(f, "\t%.*s\n", (int)reads->len[r], reads->raw + reads->qual_off[r]);
		}
		fclose(f);
		free_reads(reads);
	} else if (strcmp(argv[1], "bwtcat") == 0) {
		/* developer command (CPU only, used by the tests): .bwt -> memory through the threaded loader (load_bwt_start / blocks_ready) -> .bwt */
		if (argc < 4) { printf("Usage: bwbble bwtcat <in_bwt> <out_bwt> \n"); exit(1); }
		bwt_t *B = load_bwt_start(argv[2], 1);
		uint64_t last = 0;
		while (B->loader && last < B->num_occ) { const uint64_t r = __atomic_load_n(&B->blocks_ready, __ATOMIC_ACQUIRE); if (r < last) { printf("blocks_ready went backwards\n"); return 1; } last = r; if (r >= B->num_occ) break; }
		load_bwt_wait(B);
		if (B->blocks_ready != B->num_occ) { printf("blocks_ready %llu != %llu\n", (unsigned long long)B->blocks_ready, (unsigned long long)B->num_occ); return 1; }
		store_bwt(B, argv[3]);
		free_bwt(B);
	} else if (strcmp(argv[1], "alncat") == 0) {
		/* developer command (CPU only, used by the tests): .aln -> memory (alnsf2alns_bin) -> .aln (alns2alnf_bin read by read, or with a
		 * fifth argument `buf` the chunk serialiser of `align`, alns2alnf_buf, in chunks of a few reads) */
		if (argc < 4) { printf("Usage: bwbble alncat <in_aln> <out_aln> [buf [chunk_reads]]\n"); exit(1); }
		alns_batch_t *b = alnsf2alns_bin(argv[2]);
		FILE *f = fopen(argv[3], "wb");
		if (!f) { perror(argv[3]); return 1; }
		if (argc >= 5) {
			const size_t chunk = argc >= 6 ? (size_t)atol(argv[5]) : 1000;
			for (size_t r0 = 0; r0 < b->n_reads; r0 += chunk) {
				const size_t n = b->n_reads - r0 < chunk ? b->n_reads - r0 : chunk;
				uint64_t *off = (uint64_t *)malloc((n + 1) * 8); /* (chunk-relative, like a slot's result) */
				for (size_t k = 0; k <= n; k++) off[k] = b->aln_off[r0 + k] - b->aln_off[r0];
				size_t len = 0;
				unsigned char *buf = alns2alnf_buf(b->alns + b->aln_off[r0], off, (uint32_t)n, &len);
				fwrite(buf, 1, len, f);
				free(buf); free(off);
			}
		} else for (size_t r = 0; r < b->n_reads; r++) alns2alnf_bin(b->alns + b->aln_off[r], b->aln_off[r + 1] - b->aln_off[r], f);
		fclose(f);
		free_alns_batch(b);
	} else if (strcmp(argv[1], "hostbench") == 0) {
		/* developer command (CPU only; bench.py's host_pipeline keys): the two host stages of `align` on their own - the FASTQ reader
		 * (parallel record scan + base encoding, reads.c) in chunks of BWB_CHUNK reads, and the chunk serialiser (aln_io.c) on an .aln file */
		if (argc < 3) { printf("Usage: bwbble hostbench <reads_fastq> [<aln>]\n"); exit(1); }
		struct timespec t0, t1;
		const uint32_t chunk = getenv("BWB_CHUNK") ? (uint32_t)strtoul(getenv("BWB_CHUNK"), NULL, 10) : (1u << 21);
		clock_gettime(CLOCK_MONOTONIC, &t0);
		fq_stream *fs = fq_open(argv[2]);
		fq_chunk_t ch;
		uint64_t n = 0, cks = 0;
		while (fq_next_chunk(fs, chunk, &ch)) { n += ch.n; cks += ch.seq[(size_t)(ch.n - 1) * ch.stride] + ch.len[0]; free(ch.seq); free(ch.len); }
		fq_close(fs);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
		extern double fq_prof_s[3];
		printf("{\"parse_reads\": %llu, \"parse_s\": %.4f, \"parse_reads_per_s\": %.0f, \"parse_stages_s\": {\"scan\": %.3f, \"alloc\": %.3f, \"encode\": %.3f}, \"checksum\": %llu", (unsigned long long)n, dt, n / (dt > 0 ? dt : 1), fq_prof_s[0], fq_prof_s[1], fq_prof_s[2], (unsigned long long)cks);
		if (argc >= 4) {
			alns_batch_t *b = alnsf2alns_bin(argv[3]);
			double best = 1e30; size_t bytes = 0;
			for (int rep = 0; rep < 3; rep++) {
				clock_gettime(CLOCK_MONOTONIC, &t0);
				bytes = 0;
				for (size_t r0 = 0; r0 < b->n_reads; r0 += chunk) {
					const size_t nn = b->n_reads - r0 < chunk ? b->n_reads - r0 : chunk;
					uint64_t *off = (uint64_t *)malloc((nn + 1) * 8);
					for (size_t k = 0; k <= nn; k++) off[k] = b->aln_off[r0 + k] - b->aln_off[r0];
					size_t len = 0;
					unsigned char *buf = alns2alnf_buf(b->alns + b->aln_off[r0], off, (uint32_t)nn, &len);
					bytes += len;
					free(buf); free(off);
				}
				clock_gettime(CLOCK_MONOTONIC, &t1);
				dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
				if (dt < best) best = dt;
			}
			printf(", \"write_records\": %zu, \"write_bytes\": %zu, \"serialise_s\": %.4f, \"write_records_per_s\": %.0f", b->n_reads, bytes, best, b->n_reads / (best > 0 ? best : 1));
			free_alns_batch(b);
		}
		printf("}\n");
	} else {
		printf("Error: Unknown command '%s'\n", argv[1]);
		usage();
	}
	return 0;
}
