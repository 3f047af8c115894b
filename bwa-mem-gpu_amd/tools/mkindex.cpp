// mkindex -- from-scratch builder of a stock-format `bwa index` file set (tooling, not hot path).
//
// Writes <prefix>.pac/.ann/.amb/.bwt/.sa byte-identical to what the reference's `bwa index` writes
// for the same FASTA (checked in tests/test_oracle_golden.py and tests/test_oracle_vs_ref.py against the reference's
// own output), so indexes built here and indexes built by stock bwa are interchangeable:
//   .pac  forward strand 2-bit, N -> lrand48()&3 after srand48(11)          (bntseq.c:229-333)
//   .ann/.amb text                                                          (bntseq.c:65-94)
//   .bwt  primary, L2[1..4], Occ-interleaved BWT of  fwd + revcomp          (bwtindex.c:64-172, bwt.c:385)
//   .sa   every 32nd suffix-array value                                      (bwt.c:62-84, 396-407)
// The suffix array is built bucket by bucket (first 8 bases): a counting pass and a write-combined scatter pass
// distribute the suffixes, then every bucket is sorted on its own, in cache, by a radix sort on the next 32 bases
// (deeper comparisons only among equal keys) and turned into its BWT slice and sampled-SA entries at once.  Memory:
// ~11 bytes per text symbol (hg38-scale text, 6.2 G symbols: ~70 GB); no limit on the genome size other than RAM.
//
//   mkindex <in.fa> <prefix>            (MKINDEX_VERBOSE=1: phase timings on stderr)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <ctype.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <omp.h>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_verbose = getenv("MKINDEX_VERBOSE") != nullptr;
static double g_t0 = now_s();
static void phase(const char *what) { if (g_verbose) { const double t = now_s(); fprintf(stderr, "[mkindex] %-28s %7.2f s\n", what, t - g_t0); g_t0 = t; } }

struct Contig { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

template <class T> static T *big_alloc(size_t n)          // untouched pages: first touch happens in the parallel loops
{
	void *p = mmap(nullptr, n * sizeof(T) + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (p == MAP_FAILED) { fprintf(stderr, "[mkindex] out of memory (%zu bytes)\n", n * sizeof(T)); exit(1); }
	madvise(p, n * sizeof(T) + 4096, MADV_HUGEPAGE);
	return (T*)p;
}
template <class T> static void big_free(T *p, size_t n) { if (p) munmap((void*)p, n * sizeof(T) + 4096); }

// CPUs this process may really use: the container's CFS quota (cgroup v2 cpu.max) can be far below the core count
static int usable_cpus()
{
	int n = omp_get_max_threads();
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64]; long long per = 0;
		if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
			const long long quota = atoll(q);
			const int c = (int)((quota + per - 1) / per);
			if (c >= 1 && c < n) n = c;
		}
		fclose(f);
	}
	return n;
}

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: mkindex <in.fa> <prefix>\n"); return 1; }
	if (!getenv("OMP_NUM_THREADS")) omp_set_num_threads(usable_cpus());
	if (g_verbose) fprintf(stderr, "[mkindex] %d threads\n", omp_get_max_threads());
	const std::string prefix = argv[2];
	const int fd = open(argv[1], O_RDONLY);
	struct stat st;
	if (fd < 0 || fstat(fd, &st) != 0) { perror(argv[1]); return 1; }
	const size_t flen = (size_t)st.st_size;
	const char *txt = flen ? (const char*)mmap(nullptr, flen, PROT_READ, MAP_PRIVATE, fd, 0) : "";
	if (txt == MAP_FAILED) { perror("mmap"); return 1; }

	// ---- pack the FASTA (bns_fasta2bntseq / add1, bntseq.c:232-333) ----
	std::vector<Contig> ctg;
	std::vector<Hole> holes;
	uint8_t *base = big_alloc<uint8_t>(2 * flen + 64);                  // one base per byte; forward strand, later + reverse complement
	int64_t l_pac = 0;
	srand48(11);
	{
		uint8_t lut[256];                                              // 0..3 ACGT, 4 other printable, 5 skipped (not isgraph)
		for (int c = 0; c < 256; ++c) lut[c] = isgraph(c) ? 4 : 5;
		lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1; lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;
		const char *p = txt, *end = txt + flen;
		int lasts = 0;                                                 // the previous printable character when it was ambiguous, else 0
		while (p < end) {
			const char *nl = (const char*)memchr(p, '\n', (size_t)(end - p));
			const char *le = nl ? nl : end;
			if (*p == '>') {
				const char *q = p + 1, *nm = q;
				while (q < le && !isspace((unsigned char)*q)) ++q;
				Contig c;
				c.name.assign(nm, q - nm);
				while (q < le && isspace((unsigned char)*q)) ++q;
				const char *e = le;
				while (e > q && (e[-1] == '\n' || e[-1] == '\r')) --e;
				c.anno = e > q ? std::string(q, e - q) : std::string("(null)");
				c.offset = l_pac; c.len = 0; c.n_ambs = 0;
				ctg.push_back(c);
				lasts = 0;
			} else if (!ctg.empty()) {
				Contig &cc = ctg.back();
				const int64_t n0 = l_pac;
				for (const char *s = p; s < le; ++s) {
					int c = lut[(uint8_t)*s];
					if (c == 5) continue;
					if (c == 4) {
						if (lasts == *s) ++holes.back().len;
						else { holes.push_back({ l_pac, 1, *s }); ++cc.n_ambs; }
						c = (int)(lrand48() & 3);
						lasts = *s;
					} else lasts = 0;
					base[l_pac++] = (uint8_t)c;
				}
				cc.len += (int32_t)(l_pac - n0);
			}
			p = nl ? nl + 1 : end;
		}
	}
	if (flen) munmap((void*)txt, flen);
	close(fd);
	phase("FASTA parsed");
	{   // .pac (forward only)
		const size_t pac_n = (size_t)(l_pac >> 2) + ((l_pac & 3) ? 1 : 0);
		std::vector<uint8_t> pac(pac_n, 0);
#pragma omp parallel for schedule(static)
		for (int64_t b = 0; b < (int64_t)pac_n; ++b) {
			uint8_t x = 0;
			const int64_t lo = b << 2, hi = lo + 4 < l_pac ? lo + 4 : l_pac;
			for (int64_t i = lo; i < hi; ++i) x |= base[i] << ((~i & 3) << 1);
			pac[(size_t)b] = x;
		}
		FILE *f = fopen((prefix + ".pac").c_str(), "wb");
		fwrite(pac.data(), 1, pac.size(), f);
		uint8_t ct = 0;
		if (l_pac % 4 == 0) fwrite(&ct, 1, 1, f);
		ct = (uint8_t)(l_pac % 4);
		fwrite(&ct, 1, 1, f);
		fclose(f);
	}
	{   // .ann / .amb
		FILE *f = fopen((prefix + ".ann").c_str(), "w");
		fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)ctg.size(), 11u);
		for (auto &c : ctg) {
			fprintf(f, "%d %s", 0, c.name.c_str());
			if (!c.anno.empty()) fprintf(f, " %s\n", c.anno.c_str()); else fprintf(f, "\n");
			fprintf(f, "%lld %d %d\n", (long long)c.offset, c.len, c.n_ambs);
		}
		fclose(f);
		f = fopen((prefix + ".amb").c_str(), "w");
		fprintf(f, "%lld %d %u\n", (long long)l_pac, (int)ctg.size(), (unsigned)holes.size());
		for (auto &h : holes) fprintf(f, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
		fclose(f);
	}
	phase(".pac/.ann/.amb written");

	// ---- text = forward + reverse complement, also packed 32 bases per word (first base on top) ----
	const int64_t N = l_pac * 2;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < l_pac; ++i) base[(size_t)(N - 1 - i)] = 3 - base[(size_t)i];
	const int64_t n_w = (N + 31) / 32 + 2;
	uint64_t *w = big_alloc<uint64_t>((size_t)n_w);
#pragma omp parallel for schedule(static)
	for (int64_t wi = 0; wi < n_w; ++wi) {
		uint64_t x = 0;
		const int64_t lo = wi * 32, hi = lo + 32 < N ? lo + 32 : N;
		for (int64_t i = lo; i < hi; ++i) x |= (uint64_t)base[(size_t)i] << ((~i & 31) << 1);
		w[(size_t)wi] = x;
	}
	auto key_at = [&](int64_t i) -> uint64_t {                          // 32 bases from i < N, zero padded past N
		const int sh = (int)(i & 31) << 1;
		const uint64_t a = w[(size_t)(i >> 5)];
		return sh ? (a << sh) | (w[(size_t)(i >> 5) + 1] >> (64 - sh)) : a;
	};
	auto masked_key = [&](int64_t i) -> uint64_t {                      // the same for any i (0 at and past the end)
		if (i >= N) return 0;
		uint64_t k = key_at(i);
		const int64_t rem = N - i;
		if (rem < 32) k &= ~0ull << ((32 - rem) << 1);
		return k;
	};
	phase("text packed");

	// ---- distribute the suffixes over 4^BK buckets by their first BK bases ----
	constexpr int BK = 8;
	constexpr int64_t NB = (int64_t)1 << (2 * BK);
	std::vector<int64_t> bstart((size_t)NB + 1, 0);
	uint64_t *sa_idx = big_alloc<uint64_t>((size_t)N + 8);
	{
		const int nt = omp_get_max_threads();
		std::vector<std::vector<int64_t>> cnt((size_t)nt, std::vector<int64_t>((size_t)NB, 0));
#pragma omp parallel num_threads(nt)
		{
			std::vector<int64_t> &c = cnt[(size_t)omp_get_thread_num()];
#pragma omp for schedule(static)
			for (int64_t i = 0; i < N; ++i) ++c[(size_t)(masked_key(i) >> (64 - 2 * BK))];
		}
		// bucket b of thread t starts at bstart[b] + the counts of the earlier threads: a deterministic scatter
		int64_t run = 0;
		for (int64_t b = 0; b < NB; ++b) {
			bstart[(size_t)b] = run;
			for (int t = 0; t < nt; ++t) { const int64_t v = cnt[(size_t)t][(size_t)b]; cnt[(size_t)t][(size_t)b] = run; run += v; }
		}
		bstart[(size_t)NB] = run;
		phase("buckets counted");
		// scatter through per-thread write-combining buffers (one cache line per bucket): 65 536 interleaved output
		// streams written element by element thrash the TLB; whole lines do not
		constexpr int WC = 8;
#pragma omp parallel num_threads(nt)
		{
			std::vector<int64_t> &c = cnt[(size_t)omp_get_thread_num()];
			std::vector<uint64_t> buf((size_t)NB * WC);
			std::vector<uint8_t> fill((size_t)NB, 0);
#pragma omp for schedule(static)
			for (int64_t i = 0; i < N; ++i) {
				const size_t b = (size_t)(masked_key(i) >> (64 - 2 * BK));
				uint64_t *bb = &buf[b * WC];
				bb[fill[b]++] = (uint64_t)i;
				if (fill[b] == WC) { memcpy(sa_idx + c[b], bb, WC * 8); c[b] += WC; fill[b] = 0; }
			}
			for (size_t b = 0; b < (size_t)NB; ++b) if (fill[b]) { memcpy(sa_idx + c[b], &buf[b * WC], (size_t)fill[b] * 8); c[b] += fill[b]; }
		}
	}
	phase("suffixes scattered");

	// ---- sort every bucket; emit its BWT slice and sampled-SA entries ----
	struct Ent { uint64_t key; uint64_t idx; };                         // key: the 32 bases after the bucket prefix
	auto less = [&](const Ent &a, const Ent &b) -> bool {
		if (a.key != b.key) return a.key < b.key;
		if (a.idx == b.idx) return false;
		const int64_t i = (int64_t)a.idx, j = (int64_t)b.idx;
		// the first BK + 32 bases agree (zero padded); a suffix that ends inside the compared words is the smaller one ('$' first)
		for (int64_t d = BK;; d += 32) {
			const int64_t ra = N - i - d, rb = N - j - d;
			if (ra < 32 || rb < 32) {
				const uint64_t ka = masked_key(i + d), kb = masked_key(j + d);
				if (ka != kb) return ka < kb;
				return ra < rb;
			}
			const uint64_t ka = key_at(i + d), kb = key_at(j + d);
			if (ka != kb) return ka < kb;
		}
	};
	struct Scratch { std::vector<Ent> a, b; std::vector<uint32_t> h; };
	// LSD radix sort on the 64-bit key (4 passes of 16 bits; small buckets use std::sort), then a comparison sort inside
	// every run of equal keys (repeats; rare elsewhere)
	auto sort_bucket = [&](int64_t bk, Scratch &s) {
		const int64_t lo = bstart[(size_t)bk], hi = bstart[(size_t)bk + 1];
		const size_t n = (size_t)(hi - lo);
		s.a.resize(n);
		for (size_t r = 0; r < n; ++r) { const uint64_t i = sa_idx[(size_t)lo + r]; s.a[r] = { masked_key((int64_t)i + BK), i }; }
		if (n < 8192) std::sort(s.a.begin(), s.a.end(), less);
		else {
			s.b.resize(n);
			s.h.resize(65536);
			Ent *src = s.a.data(), *dst = s.b.data();
			for (int pass = 0; pass < 4; ++pass) {
				const int sh = pass * 16;
				std::fill(s.h.begin(), s.h.end(), 0u);
				for (size_t r = 0; r < n; ++r) ++s.h[(src[r].key >> sh) & 0xffff];
				uint32_t run = 0;
				for (size_t v = 0; v < 65536; ++v) { const uint32_t t = s.h[v]; s.h[v] = run; run += t; }
				for (size_t r = 0; r < n; ++r) dst[s.h[(src[r].key >> sh) & 0xffff]++] = src[r];
				std::swap(src, dst);
			}
			// an even number of passes: the result is back in s.a (== src)
			for (size_t r = 0; r < n;) {
				size_t e = r + 1;
				while (e < n && src[e].key == src[r].key) ++e;
				if (e - r > 1) std::sort(src + r, src + e, less);
				r = e;
			}
		}
		for (size_t r = 0; r < n; ++r) sa_idx[(size_t)lo + r] = s.a[r].idx;
	};

	uint64_t primary = 0, L2[5] = { 0, 0, 0, 0, 0 };
	{
		uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(static) reduction(+:c0,c1,c2,c3)
		for (int64_t i = 0; i < N; ++i) { const uint8_t b = base[(size_t)i]; c0 += b == 0; c1 += b == 1; c2 += b == 2; c3 += b == 3; }
		L2[1] = c0; L2[2] = c0 + c1; L2[3] = c0 + c1 + c2; L2[4] = c0 + c1 + c2 + c3;
	}
	// full suffix array rows: row 0 = "$" (position N), row r+1 = sa[r]; bw = the BWT with the '$' removed
	uint8_t *bw = big_alloc<uint8_t>((size_t)N + 8);
	const uint64_t sa_intv = 32, n_sa = ((uint64_t)N + sa_intv) / sa_intv;
	std::vector<uint64_t> ssa((size_t)n_sa, 0);                          // ssa[j] = suffix of full row j*32 = sa[j*32-1]
	{
		// the row whose BWT character is '$' (suffix 0) shifts every later row by one: sort its bucket first
		int64_t rp = -1;
		Scratch s0;
		const int64_t b0 = (int64_t)(masked_key(0) >> (64 - 2 * BK));
		if (N > 0) {
			sort_bucket(b0, s0);
			for (int64_t r = bstart[(size_t)b0]; r < bstart[(size_t)b0 + 1]; ++r) if (sa_idx[(size_t)r] == 0) rp = r;
			bw[0] = base[(size_t)N - 1];                                 // row 0: the character before '$'
		}
		primary = (uint64_t)rp + 1;
#pragma omp parallel
		{
			Scratch s;
#pragma omp for schedule(dynamic, 8)
			for (int64_t b = 0; b < NB; ++b) {
				const int64_t lo = bstart[(size_t)b], hi = bstart[(size_t)b + 1];
				if (lo == hi) continue;
				if (b != b0) sort_bucket(b, s);
				for (int64_t r = lo; r < hi; ++r) {
					const uint64_t p = sa_idx[(size_t)r];
					if (r != rp) bw[(size_t)(r < rp ? r + 1 : r)] = base[p - 1];
					if (((uint64_t)r + 1) % sa_intv == 0) ssa[(size_t)(((uint64_t)r + 1) / sa_intv)] = p;
				}
			}
		}
	}
	big_free(sa_idx, (size_t)N + 8);
	big_free(w, (size_t)n_w);
	phase("buckets sorted, BWT + SA");
	{
		const uint64_t n_blk = (uint64_t)(N + 127) / 128, n_occ = n_blk + 1;
		const uint64_t bwt_words = (uint64_t)(N + 15) / 16 + n_occ * 8;
		uint32_t *buf = big_alloc<uint32_t>((size_t)bwt_words);
		std::vector<uint64_t> cnt((size_t)(n_blk + 1) * 4, 0);           // counts before each 128-base block
#pragma omp parallel for schedule(static)
		for (int64_t b = 0; b < (int64_t)n_blk; ++b) {
			uint64_t c[4] = { 0, 0, 0, 0 };
			const int64_t lo = b * 128, hi = lo + 128 < N ? lo + 128 : N;
			for (int64_t i = lo; i < hi; ++i) ++c[bw[(size_t)i]];
			memcpy(&cnt[(size_t)(b + 1) * 4], c, 32);
		}
		for (uint64_t b = 1; b <= n_blk; ++b) for (int t = 0; t < 4; ++t) cnt[b * 4 + t] += cnt[(b - 1) * 4 + t];
#pragma omp parallel for schedule(static)
		for (int64_t b = 0; b < (int64_t)n_blk; ++b) {                   // block b: 8 words of counts, then up to 8 words of bases
			const int64_t lo = b * 128, hi = lo + 128 < N ? lo + 128 : N;
			const uint64_t k = (uint64_t)b * 16;                         // every earlier block is full (only the last can be short)
			uint32_t blk[16];
			memcpy(blk, &cnt[(size_t)b * 4], 32);
			memset(blk + 8, 0, 32);
			for (int64_t i = lo; i < hi; ++i) blk[8 + ((i - lo) >> 4)] |= (uint32_t)bw[(size_t)i] << ((~i & 15) << 1);
			const int n_words = 8 + (int)((hi - lo + 15) >> 4);
			memcpy(&buf[(size_t)k], blk, (size_t)n_words * 4);
		}
		{   // the trailing count block (bwtindex.c:169)
			const uint64_t k = (uint64_t)(N + 15) / 16 + n_blk * 8;
			memcpy(&buf[(size_t)k], &cnt[(size_t)n_blk * 4], 32);
		}
		FILE *f = fopen((prefix + ".bwt").c_str(), "wb");
		fwrite(&primary, 8, 1, f); fwrite(L2 + 1, 8, 4, f);
		fwrite(buf, 4, (size_t)bwt_words, f);
		fclose(f);
		big_free(buf, (size_t)bwt_words);
	}
	phase(".bwt written");
	{
		const uint64_t seq_len = (uint64_t)N;
		FILE *f = fopen((prefix + ".sa").c_str(), "wb");
		fwrite(&primary, 8, 1, f); fwrite(L2 + 1, 8, 4, f);
		fwrite(&sa_intv, 8, 1, f); fwrite(&seq_len, 8, 1, f);
		fwrite(ssa.data() + 1, 8, (size_t)n_sa - 1, f);
		fclose(f);
	}
	phase(".sa written");
	return 0;
}
