/* aln_io.c - .aln records, byte-compatible with alns2alnf_bin / alnsf2alns_bin (mg-aligner/align.c:345-382,
 * 430-483): i32 num_entries, then per entry i32 score, u64 L, u64 U, i32 num_mm, num_gapo,
 * num_gape, aln_length, i32 state_pairs, i32 pair[] with pair = state | (count << 2), path walked from index
 * aln_length-1 down to 0. */
#include <stdlib.h>
#include <string.h>
#include "bwb_host.h"

/* the edit path is all STATE_M(0) except for the gap runs recorded by the kernel */
int aln_path_bytes(const bwb_aln *a, unsigned char *path) {
	memset(path, 0, 272);
	for (int k = 0; k < BWB_MAX_GAP_RUNS; k++) {
		const unsigned run = a->gap_run[k];
		if (run == 0xFFFFu) continue;
		const unsigned start = run & 0xFF, len = (run >> 8) & 0x7F, st = (run >> 15) ? 2 : 1; /* STATE_D : STATE_I */
		for (unsigned q = 0; q < len && start + q < 272; q++) path[start + q] = (unsigned char)st;
	}
	return (int)a->aln_length;
}

void alns2alnf_bin(const bwb_aln *alns, uint64_t n, FILE *f) {
	int32_t buf[8 + 272];
	int32_t ne = (int32_t)n;
	fwrite(&ne, 4, 1, f);
	unsigned char path[272];
	for (uint64_t i = 0; i < n; i++) {
		const bwb_aln *a = &alns[i];
		const int alen = aln_path_bytes(a, path);
		int32_t score = a->score, v[4] = { a->num_mm, a->num_gapo, a->num_gape, alen };
		fwrite(&score, 4, 1, f);
		fwrite(&a->L, 8, 1, f);
		fwrite(&a->U, 8, 1, f);
		fwrite(v, 4, 4, f);
		int32_t pairs = 0;
		if (alen > 0) {
			int state = path[alen - 1];
			uint16_t counter = 1;
			pairs = 1;
			for (int j = alen - 2; j >= 0; j--) {
				if (state == path[j]) counter++;
				else { buf[pairs] = state | (counter << 2); state = path[j]; counter = 1; pairs++; }
			}
			buf[pairs] = state | (counter << 2);
			buf[0] = pairs;
			fwrite(buf, 4, (size_t)pairs + 1, f);
		} else fwrite(&pairs, 4, 1, f);
	}
}

/* The records of a whole chunk of reads as ONE byte buffer (round 5: built by the GPU worker that received the chunk's hits, with all
 * cores, so that the ordered writer only write()s; per-read fwrites in the writer thread were what a stream of several GPUs queued
 * behind).  Same bytes as alns2alnf_bin read by read.  Returns the malloc'ed buffer, *len = its length. */
static inline int aln_nogap(const bwb_aln *a) { uint64_t r[2]; memcpy(r, a->gap_run, 16); return (r[0] & r[1]) == ~0ull; }
static inline size_t aln_rec_bytes(const bwb_aln *a) {
	/* 36 bytes of fixed fields + the pair count + one pair per run of equal states in the path */
	if (a->aln_length == 0) return 40;
	unsigned char path[272];
	const int alen = aln_path_bytes(a, path);
	int pairs = 1;
	for (int j = alen - 2; j >= 0; j--) pairs += path[j] != path[j + 1];
	return 40 + 4 * (size_t)pairs;
}
static inline unsigned char *aln_rec_put(const bwb_aln *a, unsigned char *o) {
	int32_t hdr[9];
	unsigned char path[272];
	const int nogap = aln_nogap(a);
	const int alen = nogap ? (int)a->aln_length : aln_path_bytes(a, path);
	hdr[0] = a->score; memcpy(hdr + 1, &a->L, 8); memcpy(hdr + 3, &a->U, 8);
	hdr[5] = a->num_mm; hdr[6] = a->num_gapo; hdr[7] = a->num_gape; hdr[8] = alen;
	memcpy(o, hdr, 36); o += 36;
	if (alen <= 0) { const int32_t z = 0; memcpy(o, &z, 4); return o + 4; }
	if (nogap) { const int32_t v[2] = { 1, 0 | (alen << 2) }; memcpy(o, v, 8); return o + 8; } /* (all STATE_M: one pair) */
	int32_t *pp = (int32_t *)o; /* (4-byte aligned: every field before it is) */
	int pairs = 0, state = path[alen - 1];
	uint16_t counter = 1;
	for (int j = alen - 2; j >= 0; j--) {
		if (state == path[j]) counter++;
		else { pp[++pairs] = state | (counter << 2); state = path[j]; counter = 1; }
	}
	pp[++pairs] = state | (counter << 2);
	pp[0] = pairs;
	return o + 4 * ((size_t)pairs + 1);
}
unsigned char *alns2alnf_buf(const bwb_aln *alns, const uint64_t *aln_off, uint32_t n_reads, size_t *len) {
	size_t *pos = (size_t *)malloc(((size_t)n_reads + 1) * sizeof(size_t));
#pragma omp parallel for schedule(static) num_threads(bwb_host_team())
	for (long r = 0; r < (long)n_reads; r++) {
		size_t b = 4;
		for (uint64_t i = aln_off[r]; i < aln_off[r + 1]; i++) {
			const bwb_aln *a = &alns[i];
			const int nogap = aln_nogap(a);
			b += (nogap && a->aln_length) ? 44 : aln_rec_bytes(a);
		}
		pos[r + 1] = b;
	}
	pos[0] = 0;
	for (uint32_t r = 0; r < n_reads; r++) pos[r + 1] += pos[r];
	unsigned char *buf = (unsigned char *)malloc(pos[n_reads] ? pos[n_reads] : 1);
#pragma omp parallel for schedule(static) num_threads(bwb_host_team())
	for (long r = 0; r < (long)n_reads; r++) {
		unsigned char *o = buf + pos[r];
		const int32_t ne = (int32_t)(aln_off[r + 1] - aln_off[r]);
		memcpy(o, &ne, 4); o += 4;
		for (uint64_t i = aln_off[r]; i < aln_off[r + 1]; i++) o = aln_rec_put(&alns[i], o);
	}
	*len = pos[n_reads];
	free(pos);
	return buf;
}

alns_batch_t *alnsf2alns_bin(const char *alnFname) {
	FILE *f = fopen(alnFname, "rb");
	if (!f) bwb_die("alnsf2alns: Cannot open ALN file: %s!", alnFname);
	alns_batch_t *b = (alns_batch_t *)calloc(1, sizeof(*b));
	size_t rcap = 1u << 16, acap = 1u << 16, na = 0;
	b->aln_off = (uint64_t *)malloc((rcap + 1) * 8);
	b->alns = (bwb_aln *)malloc(acap * sizeof(bwb_aln));
	b->aln_off[0] = 0;
	int32_t ne;
	while (fread(&ne, 4, 1, f) == 1) {
		if (b->n_reads == rcap) { rcap *= 2; b->aln_off = (uint64_t *)realloc(b->aln_off, (rcap + 1) * 8); }
		for (int i = 0; i < ne; i++) {
			if (na == acap) { acap *= 2; b->alns = (bwb_aln *)realloc(b->alns, acap * sizeof(bwb_aln)); }
			bwb_aln *a = &b->alns[na++];
			int32_t score, v[4], pairs;
			if (fread(&score, 4, 1, f) < 1 || fread(&a->L, 8, 1, f) < 1 || fread(&a->U, 8, 1, f) < 1 || fread(v, 4, 4, f) < 4 || fread(&pairs, 4, 1, f) < 1)
				bwb_die("alnsf2alns: Could not read ALN file: %s!", alnFname);
			/* (aln_length comes from the file: the record builders index a 272-byte path with it - aln_rec_put / aln_rec_bytes -, and the
			 * reference's own field is 8 bits wide, align.h:103: anything beyond 255, a negative count of entries or of pairs is a broken file) */
			if (v[3] < 0 || v[3] > 255 || pairs < 0 || pairs > 256) bwb_die("alnsf2alns: %s: a record with aln_length %d and %d state pairs is not an ALN record", alnFname, v[3], pairs);
			a->score = (uint16_t)score; a->num_mm = (uint8_t)v[0]; a->num_gapo = (uint8_t)v[1]; a->num_gape = (uint8_t)v[2];
			a->aln_length = (uint16_t)v[3]; a->reserved = 0;
			for (int k = 0; k < BWB_MAX_GAP_RUNS; k++) a->gap_run[k] = 0xFFFF;
			a->reserved2 = 0;
			/* The reference loader fills aln_path in PAIR order (align.c:466-476), i.e. the loaded path is the
			 * align-time path reversed (pairs were written from index aln_length-1 down to 0, align.c:363-373);
			 * eval_aln / print_aln2sam work on that orientation, so the gap runs are rebuilt the same way. */
			int32_t *pp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1));
			if (pairs > 0 && fread(pp, 4, (size_t)pairs, f) < (size_t)pairs) bwb_die("alnsf2alns: Could not read ALN file: %s!", alnFname);
			int pos = 0, nrun = 0;
			for (int k = 0; k < pairs; k++) {
				const int st = pp[k] & 3, cnt = pp[k] >> 2;
				if (st != 0 && nrun < BWB_MAX_GAP_RUNS) a->gap_run[nrun++] = (uint16_t)((pos & 0xFF) | ((cnt & 0x7F) << 8) | (st == 2 ? 0x8000 : 0));
				pos += cnt;
			}
			free(pp);
		}
		b->n_reads++;
		b->aln_off[b->n_reads] = na;
	}
	fclose(f);
	return b;
}

void free_alns_batch(alns_batch_t *b) {
	if (!b) return;
	free(b->aln_off); free(b->alns); free(b);
}
