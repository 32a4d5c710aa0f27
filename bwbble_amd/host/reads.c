/* reads.c - FASTQ reader with the reference's encoding (mg-aligner/io.c:410-515, tables io.h:108-130).
 * The reference mallocs three buffers per read; here the file is read once and the codes go into one
 * [count][stride] array that is handed to the GPU as is. */
#include <stdlib.h>
#include <string.h>
#include "bwb_host.h"

static inline uint8_t nt4(char c) { /* nt4_table io.h:113-130: A0 G1 C2 T3, everything else 4 */
	switch (c) {
	case 'A': case 'a': return 0;
	case 'G': case 'g': return 1;
	case 'C': case 'c': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

reads_t *fastq2reads(const char *readsFname) {
	FILE *f = fopen(readsFname, "r");
	if (!f) bwb_die("load_reads_fastq: Cannot open reads file: %s !", readsFname);
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	reads_t *R = (reads_t *)calloc(1, sizeof(reads_t));
	R->raw = (char *)malloc((size_t)sz + 1);
	if (!R->raw || fread(R->raw, 1, (size_t)sz, f) != (size_t)sz) bwb_die("load_reads_fastq: Cannot read reads file: %s !", readsFname);
	fclose(f);
	R->raw[sz] = 0;
	const char *raw = R->raw;
	size_t cap = 1u << 16, n = 0;
	size_t *soff = (size_t *)malloc(cap * sizeof(size_t));
	R->name_off = (size_t *)malloc(cap * sizeof(size_t));
	R->qual_off = (size_t *)malloc(cap * sizeof(size_t));
	R->name_len = (uint16_t *)malloc(cap * sizeof(uint16_t));
	R->len = (uint16_t *)malloc(cap * sizeof(uint16_t));
	long p = 0;
	for (;;) {
		while (p < sz && raw[p] != '@') p++;                 /* io.c:430-434 */
		if (p >= sz) break;
		p++;
		const long ns = p;
		while (p < sz && raw[p] != '\n') p++;                /* line 1 */
		if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", readsFname);
		long nl = p - ns;
		if (nl > MAX_SEQ_NAME_LEN) nl = MAX_SEQ_NAME_LEN;    /* io.c:439 */
		p++;
		const long ss = p;
		while (p < sz && raw[p] != '\n') p++;                /* line 2 */
		if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", readsFname);
		const long sl = p - ss;
		while (p < sz && raw[p] != '+') p++;                 /* io.c:474-477 */
		while (p < sz && raw[p] != '\n') p++;                /* line 3 */
		if (p >= sz) bwb_die("Error: The input file %s is not in the FASTQ format.", readsFname);
		p++;
		const long qs = p;
		while (p < sz && raw[p] != '\n') p++;                /* line 4 */
		if (p - qs != sl) bwb_die("Error: The number of quality score symbols does not match the length of the read sequence."); /* io.c:495-498 */
		if (sl > 65535) bwb_die("Error: read longer than 65535 bases.");
		if (n == cap) {
			cap *= 2;
			soff = (size_t *)realloc(soff, cap * sizeof(size_t));
			R->name_off = (size_t *)realloc(R->name_off, cap * sizeof(size_t));
			R->qual_off = (size_t *)realloc(R->qual_off, cap * sizeof(size_t));
			R->name_len = (uint16_t *)realloc(R->name_len, cap * sizeof(uint16_t));
			R->len = (uint16_t *)realloc(R->len, cap * sizeof(uint16_t));
		}
		soff[n] = (size_t)ss; R->name_off[n] = (size_t)ns; R->name_len[n] = (uint16_t)nl; R->qual_off[n] = (size_t)qs;
		R->len[n] = (uint16_t)sl;
		if ((unsigned)sl > R->max_len) R->max_len = (unsigned)sl;
		n++;
	}
	R->count = (unsigned)n;
	R->stride = R->max_len ? R->max_len : 1;
	R->seq = (uint8_t *)malloc((n ? n : 1) * (size_t)R->stride);
	memset(R->seq, 4, (n ? n : 1) * (size_t)R->stride);
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)n; i++) {
		uint8_t *d = R->seq + (size_t)i * R->stride;
		const char *s = raw + soff[i];
		for (int k = 0; k < R->len[i]; k++) d[k] = nt4(s[k]);   /* io.c:467 */
	}
	free(soff);
	printf("Loaded %d reads from %s.\n", R->count, readsFname);
	return R;
}

void free_reads(reads_t *R) {
	if (!R) return;
	free(R->seq); free(R->len); free(R->raw); free(R->name_off); free(R->qual_off); free(R->name_len); free(R);
}
