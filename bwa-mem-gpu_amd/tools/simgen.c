/* simgen -- deterministic synthetic genomes and read sets (tooling, not hot path).
 *
 * Implements the synthetic inputs SURVEY.md section 8(d) / BASELINE.md section 3
 * prescribe: i.i.d. ACGT genome (optionally with dispersed/tandem repeat
 * families and N holes so the occ>10 / >20 / >500 seeding paths are exercised),
 * reads with uniform starts, 50 % reverse-complement, per-base substitution
 * error to a *different* base, optional indels / N's, constant quality 'I',
 * names r<index>; PE: fragment ~ round(N(500,50^2)) clipped to [300,700],
 * mate 2 = reverse complement of the fragment end (FR), same name on both mates.
 *
 * Everything is driven by splitmix64 so the output depends only on the seed.
 *
 *   simgen genome <out.fa> <seed> <repeat_mode 0|1> <len1> [len2 ...]
 *   simgen reads  <in.fa> <out1.fq> <out2.fq|-> <n_reads> <read_len> <sub_ppm> <indel_ppm> <n_ppm> <seed> [chim_ppm]
 *                 (n_reads counts reads, so PE writes n_reads/2 pairs; out2 "-" = SE)
 *
 * Also a library: simgen_* functions below are exported for ctypes (bench.py
 * fills read batches directly in memory).
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t s; } sg_rng;
static inline uint64_t sg_next(sg_rng *r)
{
	uint64_t z = (r->s += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}
static inline uint64_t sg_below(sg_rng *r, uint64_t n) { return (uint64_t)(((__uint128_t)sg_next(r) * n) >> 64); }
static inline double sg_unif(sg_rng *r) { return (sg_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static double sg_normal(sg_rng *r)
{
	double u1 = sg_unif(r), u2 = sg_unif(r);
	if (u1 < 1e-300) u1 = 1e-300;
	return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* Fill seq[0..len) with i.i.d. A/C/G/T as ASCII. */
void simgen_random_bases(uint64_t seed, int64_t len, char *seq)
{
	sg_rng r = { seed };
	int64_t i;
	for (i = 0; i + 32 <= len; i += 32) {
		uint64_t x = sg_next(&r);
		int j;
		for (j = 0; j < 32; ++j, x >>= 2) seq[i + j] = "ACGT"[x & 3];
	}
	for (; i < len; ++i) seq[i] = "ACGT"[sg_next(&r) & 3];
}

/* Inject repeat families, tandem repeats and N holes into one contig. */
void simgen_add_repeats(uint64_t seed, int64_t len, char *seq)
{
	sg_rng r = { seed ^ 0x5bd1e995u };
	int fam;
	if (len < 20000) return;
	/* dispersed families: (unit length, copies, divergence ppm) */
	static const int fams[][3] = {
		{ 300, 12, 20000 }, { 300, 30, 50000 }, { 120, 700, 10000 }, { 1000, 6, 5000 },
		{ 60, 25, 0 }, { 2000, 3, 0 }, { 150, 40, 30000 }, { 40, 600, 0 },
	};
	for (fam = 0; fam < (int)(sizeof(fams) / sizeof(fams[0])); ++fam) {
		int L = fams[fam][0], K = fams[fam][1], div = fams[fam][2], k, i;
		int64_t scale = len / 1000000 > 0 ? len / 1000000 : 1;
		int64_t src = sg_below(&r, len - L);
		if (scale > 8) scale = 8;
		for (k = 0; k < K * (int)(scale > 2 ? 2 : scale); ++k) {
			int64_t dst = sg_below(&r, len - L);
			int rc = sg_next(&r) & 1;
			if (dst + L > src && dst < src + L) continue;
			for (i = 0; i < L; ++i) {
				char c = rc ? seq[src + L - 1 - i] : seq[src + i];
				if (rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
				if (div && sg_below(&r, 1000000) < (uint64_t)div) c = "ACGT"[sg_next(&r) & 3];
				seq[dst + i] = c;
			}
		}
	}
	/* tandem repeats */
	for (fam = 0; fam < 6; ++fam) {
		int unit = 2 + (int)sg_below(&r, 30), n = 10 + (int)sg_below(&r, 40), i;
		int64_t at = sg_below(&r, len - (int64_t)unit * n - 1);
		for (i = unit; i < unit * n; ++i) seq[at + i] = seq[at + i % unit];
	}
	/* N holes */
	for (fam = 0; fam < 4; ++fam) {
		int L = 1 + (int)sg_below(&r, fam == 0 ? 1 : 500), i;
		int64_t at = sg_below(&r, len - L);
		for (i = 0; i < L; ++i) seq[at + i] = 'N';
	}
}

/* "human-like" repeat load (bench.py --genome-profile human-like): what GRCh38 brings and the default profile lacks -- a large share of
 * the bases inside dispersed families with 10^4..10^6 copies genome-wide, every copy 5..15 % diverged from its family's consensus, plus
 * satellite arrays.  The consensus sequences depend on `family_seed` only, so every contig of a genome carries the SAME families (the copy
 * numbers below are per base of contig: a 3.1 Gbp genome gets ~1.2 M Alu-like, ~0.9 M L1-like fragments, 20 x ~10^4 mid-frequency copies;
 * 15 % of the Alu- and L1-like copies are young ones at 1..5 %):
 *   Alu-like   8 subfamilies (300 bp, 2-4 % apart), 11 % of the bases, full length copies;
 *   L1-like    4 subfamilies (6 000 bp), 26 % of the bases, 5'-truncated copies (the 3' end, length ~ 100 + Exp(900), at most 6 000);
 *   mid        20 families of 500 bp, 0.16 % of the bases each (~10^4 copies per 3.1 Gbp);
 *   satellite  two arrays per contig of a 171 bp unit (alpha-satellite-like), 1.5 % of the contig each, copies 1-3 % diverged;
 * then the default profile's tandem repeats and N holes.  Insertions overwrite what is there (nested repeats, as in a real genome). */
static void sg_consensus(uint64_t seed, int len, char *out) { simgen_random_bases(seed, len, out); }
static void sg_insert_copy(sg_rng *r, char *seq, int64_t dst, const char *cons, int L, int div_ppm, int rc)
{
	int i;
	for (i = 0; i < L; ++i) {
		char c = rc ? cons[L - 1 - i] : cons[i];
		if (rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
		if (sg_below(r, 1000000) < (uint64_t)div_ppm) {
			uint64_t u = sg_below(r, 20);
			if (u == 0) continue;                                   /* 1 in 20 differences is a deleted base (the copy gets shorter) */
			c = "ACGT"[sg_next(r) & 3];
		}
		seq[dst++] = c;
	}
}
/* divergence of one copy from its consensus: 85 % of the copies 5-15 %, 15 % young copies at 1-5 % */
static int sg_div(sg_rng *r) { return sg_below(r, 100) < 15 ? 10000 + (int)sg_below(r, 40000) : 50000 + (int)sg_below(r, 100000); }
void simgen_add_human_repeats(uint64_t family_seed, uint64_t seed, int64_t len, char *seq)
{
	enum { N_ALU = 8, L_ALU = 300, N_L1 = 4, L_L1 = 6000, N_MID = 20, L_MID = 500, L_SAT = 171 };
	static char alu[N_ALU][L_ALU], l1[N_L1][L_L1], mid[N_MID][L_MID], sat[2][L_SAT];
	sg_rng fr = { family_seed ^ 0xa17a17a17ULL }, r = { seed ^ 0x2545f491ULL };
	int64_t k, n;
	int i, f;
	if (len < 100000) return;
	/* consensus sequences (the same for every contig) */
	sg_consensus(family_seed + 11, L_ALU, alu[0]);
	for (f = 1; f < N_ALU; ++f) { memcpy(alu[f], alu[0], L_ALU); for (i = 0; i < L_ALU; ++i) if (sg_below(&fr, 100) < 3) alu[f][i] = "ACGT"[sg_next(&fr) & 3]; }
	sg_consensus(family_seed + 13, L_L1, l1[0]);
	for (f = 1; f < N_L1; ++f) { memcpy(l1[f], l1[0], L_L1); for (i = 0; i < L_L1; ++i) if (sg_below(&fr, 100) < 4) l1[f][i] = "ACGT"[sg_next(&fr) & 3]; }
	for (f = 0; f < N_MID; ++f) sg_consensus(family_seed + 100 + f, L_MID, mid[f]);
	for (f = 0; f < 2; ++f) sg_consensus(family_seed + 200 + f, L_SAT, sat[f]);
	/* L1-like first (oldest, longest), then mid, then Alu-like (youngest: lands inside the others too) */
	for (n = 0; n < (int64_t)(0.26 * len); ) {
		int L = 100 + (int)(-900.0 * log(1.0 - sg_unif(&r)));
		if (L > L_L1) L = L_L1;
		f = (int)sg_below(&r, N_L1);
		sg_insert_copy(&r, seq, (int64_t)sg_below(&r, len - L), l1[f] + (L_L1 - L), L, sg_div(&r), (int)(sg_next(&r) & 1));
		n += L;
	}
	for (f = 0; f < N_MID; ++f)
		for (k = 0; k < (int64_t)(0.0016 * len / L_MID); ++k)
			sg_insert_copy(&r, seq, (int64_t)sg_below(&r, len - L_MID), mid[f], L_MID, 50000 + (int)sg_below(&r, 100000), (int)(sg_next(&r) & 1));
	for (k = 0; k < (int64_t)(0.11 * len / L_ALU); ++k)
		sg_insert_copy(&r, seq, (int64_t)sg_below(&r, len - L_ALU), alu[sg_below(&r, N_ALU)], L_ALU, sg_div(&r), (int)(sg_next(&r) & 1));
	for (f = 0; f < 2; ++f) {
		int64_t span = (int64_t)(0.015 * len) / L_SAT * L_SAT, at = (int64_t)sg_below(&r, len - span - L_SAT);
		for (k = 0; k < span; k += L_SAT) sg_insert_copy(&r, seq, at + k, sat[f], L_SAT, 10000 + (int)sg_below(&r, 20000), 0);
	}
	/* tandem repeats and N holes as in the default profile */
	for (f = 0; f < 6; ++f) {
		int unit = 2 + (int)sg_below(&r, 30), m = 10 + (int)sg_below(&r, 40);
		int64_t at = sg_below(&r, len - (int64_t)unit * m - 1);
		for (i = unit; i < unit * m; ++i) seq[at + i] = seq[at + i % unit];
	}
	for (f = 0; f < 4; ++f) {
		int L = 1 + (int)sg_below(&r, f == 0 ? 1 : 500);
		int64_t at = sg_below(&r, len - L);
		for (i = 0; i < L; ++i) seq[at + i] = 'N';
	}
}

static inline char sg_comp(char c)
{
	switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

/* Draw one read of length rl from the fragment frag[0..fl) (already oriented),
 * applying substitution / indel / N errors.  Output is exactly rl bases. */
static void sg_mutate(sg_rng *r, const char *frag, int fl, int rl, int sub_ppm, int indel_ppm, int n_ppm, char *out)
{
	int i = 0, j = 0;
	while (j < rl) {
		uint64_t u = sg_below(r, 1000000);
		char c = i < fl ? frag[i] : "ACGT"[sg_next(r) & 3];
		if (u < (uint64_t)indel_ppm) {
			int L = 1 + (int)sg_below(r, 3);
			if (sg_next(r) & 1) { i += L; continue; } /* deletion from the read */
			while (L-- > 0 && j < rl) out[j++] = "ACGT"[sg_next(r) & 3]; /* insertion */
			continue;
		}
		if (u < (uint64_t)indel_ppm + sub_ppm && c != 'N') {
			int k = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
			c = "ACGT"[(k + 1 + sg_below(r, 3)) & 3];
		} else if (u < (uint64_t)indel_ppm + sub_ppm + n_ppm) c = 'N';
		out[j++] = c; ++i;
	}
}

/* Generate reads in memory.  genome: concatenated ASCII contigs (total glen),
 * contig boundaries in off[0..n_ctg] (off[n_ctg] = glen).  Reads never span a
 * contig boundary (except chimeric ones, which join two loci).  SE: n reads.
 * PE (is_pe): n must be even; reads 2i / 2i+1 are mates.  out: n * rl bytes. */
void simgen_reads(uint64_t seed, const char *genome, int n_ctg, const int64_t *off, int64_t n, int rl,
                  int sub_ppm, int indel_ppm, int n_ppm, int chim_ppm, int is_pe, char *out)
{
	sg_rng r = { seed };
	int64_t glen = off[n_ctg], i;
	int maxfrag = is_pe ? 700 : rl + 64;
	char *frag = (char*)malloc(maxfrag + 8), *tmp = (char*)malloc(maxfrag + 8);
	for (i = 0; i < n; i += is_pe ? 2 : 1) {
		int fl = rl + (indel_ppm ? 16 : 0), k, c, rev;
		int64_t pos;
		if (is_pe) {
			fl = (int)floor(500.0 + 50.0 * sg_normal(&r) + 0.5);
			if (fl < 300) fl = 300;
			if (fl > 700) fl = 700;
			if (fl < rl) fl = rl;
		}
		for (;;) { /* uniform start over windows that fit inside one contig */
			pos = sg_below(&r, glen);
			for (c = 0; c < n_ctg && off[c + 1] <= pos; ++c);
			if (pos + fl <= off[c + 1]) break;
			if (off[c + 1] - off[c] < fl) { fl = (int)(off[c + 1] - off[c]); pos = off[c]; if (fl >= rl) break; }
		}
		rev = sg_next(&r) & 1;
		for (k = 0; k < fl; ++k) frag[k] = rev ? sg_comp(genome[pos + fl - 1 - k]) : genome[pos + k];
		if (!is_pe && chim_ppm && sg_below(&r, 1000000) < (uint64_t)chim_ppm) { /* chimeric: second half from elsewhere */
			int cut = rl / 3 + (int)sg_below(&r, rl / 3);
			int64_t p2 = sg_below(&r, glen - rl);
			int rev2 = sg_next(&r) & 1;
			for (k = cut; k < fl; ++k) frag[k] = rev2 ? sg_comp(genome[p2 + fl - 1 - k]) : genome[p2 + k];
		}
		sg_mutate(&r, frag, fl, rl, sub_ppm, indel_ppm, n_ppm, out + i * rl);
		if (is_pe) {
			for (k = 0; k < fl; ++k) tmp[k] = sg_comp(frag[fl - 1 - k]);
			sg_mutate(&r, tmp, fl, rl, sub_ppm, indel_ppm, n_ppm, out + (i + 1) * rl);
		}
	}
	free(frag); free(tmp);
}

#ifndef SIMGEN_NO_MAIN
static char *read_fasta(const char *fn, int *n_ctg, int64_t **off_)
{
	FILE *fp = fopen(fn, "r");
	int64_t m = 1 << 20, l = 0, *off = (int64_t*)malloc(8 * 65536);
	char *g = (char*)malloc(m), line[1 << 16];
	int nc = 0;
	if (!fp) { perror(fn); exit(1); }
	while (fgets(line, sizeof line, fp)) {
		if (line[0] == '>') { off[nc++] = l; continue; }
		int k = strlen(line);
		while (k && (line[k-1] == '\n' || line[k-1] == '\r')) --k;
		if (l + k > m) { while (l + k > m) m <<= 1; g = (char*)realloc(g, m); }
		for (int i = 0; i < k; ++i) { char c = line[i] & ~0x20; g[l++] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N'; }
	}
	off[nc] = l;
	fclose(fp);
	*n_ctg = nc; *off_ = off;
	return g;
}

int main(int argc, char **argv)
{
	if (argc >= 6 && strcmp(argv[1], "genome") == 0) {
		FILE *fp = fopen(argv[2], "w");
		uint64_t seed = strtoull(argv[3], 0, 10);
		int rep = atoi(argv[4]), c;
		for (c = 5; c < argc; ++c) {
			int64_t len = atoll(argv[c]), i;
			char *s = (char*)malloc(len + 1);
			simgen_random_bases(seed + 1000003ULL * (c - 5), len, s);
			if (rep) simgen_add_repeats(seed + 7919ULL * (c - 5), len, s);
			fprintf(fp, ">ctg%d\n", c - 4);
			for (i = 0; i < len; i += 60) { fwrite(s + i, 1, len - i < 60 ? len - i : 60, fp); fputc('\n', fp); }
			free(s);
		}
		fclose(fp);
		return 0;
	}
	if (argc >= 11 && strcmp(argv[1], "reads") == 0) {
		int n_ctg, is_pe = strcmp(argv[4], "-") != 0, rl = atoi(argv[6]);
		int64_t *off, n = atoll(argv[5]), i;
		char *g = read_fasta(argv[2], &n_ctg, &off);
		char *out = (char*)malloc(n * rl);
		FILE *f1 = fopen(argv[3], "w"), *f2 = is_pe ? fopen(argv[4], "w") : 0;
		simgen_reads(strtoull(argv[10], 0, 10), g, n_ctg, off, n, rl, atoi(argv[7]), atoi(argv[8]), atoi(argv[9]),
		             argc > 11 ? atoi(argv[11]) : 0, is_pe, out);
		for (i = 0; i < n; ++i) {
			FILE *fp = is_pe && (i & 1) ? f2 : f1;
			int k;
			fprintf(fp, "@r%lld\n", (long long)(is_pe ? i >> 1 : i));
			fwrite(out + i * rl, 1, rl, fp);
			fputs("\n+\n", fp);
			for (k = 0; k < rl; ++k) fputc('I', fp);
			fputc('\n', fp);
		}
		fclose(f1); if (f2) fclose(f2);
		return 0;
	}
	fprintf(stderr, "usage: simgen genome <out.fa> <seed> <repeat_mode> <len>...\n"
	                "       simgen reads <in.fa> <out1.fq> <out2.fq|-> <n> <len> <sub_ppm> <indel_ppm> <n_ppm> <seed> [chim_ppm]\n");
	return 1;
}
#endif
