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
