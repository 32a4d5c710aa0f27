/* bwt_io.c - .bwt file I/O, byte-compatible with the reference (mg-aligner/bwt.c:66-125):
 * u64 length, num_words, num_sa, num_occ, sa0_index, u64 C[17], u32 bwt[num_words], u64 O[num_occ*16], u64 SA[num_sa]. */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "bwb_host.h"

void bwb_die(const char *fmt, ...) { /* the reference printf()s and exit(1)s on every error (e.g. bwt.c:68-70) */
	va_list ap;
	va_start(ap, fmt);
	vprintf(fmt, ap);
	va_end(ap);
	printf("\n");
	exit(1);
}

void store_bwt(const bwt_t *BWT, const char *bwtFname) {
	FILE *f = fopen(bwtFname, "wb");
	if (!f) bwb_die("store_bwt: Cannot open the BWT file %s!", bwtFname);
	const bwtint_t hdr[5] = { BWT->length, BWT->num_words, BWT->num_sa, BWT->num_occ, BWT->sa0_index };
	fwrite(hdr, sizeof(bwtint_t), 5, f);
	fwrite(BWT->C, sizeof(bwtint_t), ALPHABET_SIZE + 1, f);
	fwrite(BWT->bwt, sizeof(uint32_t), BWT->num_words, f);
	fwrite(BWT->O, sizeof(bwtint_t), BWT->num_occ * ALPHABET_SIZE, f);
	fwrite(BWT->SA, sizeof(bwtint_t), BWT->num_sa, f);
	fclose(f);
}

bwt_t *load_bwt(const char *bwtFname, int loadSA) {
	FILE *f = fopen(bwtFname, "rb");
	if (!f) bwb_die("load_bwt: Cannot open the BWT file: %s!", bwtFname);
	bwt_t *B = (bwt_t *)calloc(1, sizeof(bwt_t));
	bwtint_t hdr[5];
	if (fread(hdr, sizeof(bwtint_t), 5, f) < 5) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	B->length = hdr[0]; B->num_words = hdr[1]; B->num_sa = hdr[2]; B->num_occ = hdr[3]; B->sa0_index = hdr[4];
	if (fread(B->C, sizeof(bwtint_t), ALPHABET_SIZE + 1, f) < ALPHABET_SIZE + 1) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	B->bwt = (uint32_t *)calloc(B->num_words, sizeof(uint32_t));
	B->O = (bwtint_t *)calloc(B->num_occ * ALPHABET_SIZE, sizeof(bwtint_t));
	if (!B->bwt || !B->O) bwb_die("load_bwt: Could not allocate memory for the BWT index. ");
	if (fread(B->bwt, sizeof(uint32_t), B->num_words, f) < B->num_words) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	if (fread(B->O, sizeof(bwtint_t), B->num_occ * ALPHABET_SIZE, f) < B->num_occ * ALPHABET_SIZE) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	if (loadSA) {
		B->SA = (bwtint_t *)calloc(B->num_sa, sizeof(bwtint_t));
		if (!B->SA) bwb_die("load_bwt: Could not allocate memory for the BWT index. ");
		if (fread(B->SA, sizeof(bwtint_t), B->num_sa, f) < B->num_sa) bwb_die("load_bwt: Could not read BWT from file: %s!", bwtFname);
	}
	fclose(f);
	return B;
}

void free_bwt(bwt_t *B) {
	if (!B) return;
	free(B->bwt); free(B->O); free(B->SA); free(B);
}
