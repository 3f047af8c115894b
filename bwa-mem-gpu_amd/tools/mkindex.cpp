// mkindex -- from-scratch builder of a stock-format `bwa index` file set (tooling, not hot path).
//
// Writes <prefix>.pac/.ann/.amb/.bwt/.sa byte-identical to what the reference's `bwa index` writes
// for the same FASTA (checked in tests/test_tools.py against oracle/_ref), so indexes built here and
// indexes built by stock bwa are interchangeable:
//   .pac  forward strand 2-bit, N -> lrand48()&3 after srand48(11)          (bntseq.c:229-333)
//   .ann/.amb text                                                          (bntseq.c:65-94)
//   .bwt  primary, L2[1..4], Occ-interleaved BWT of  fwd + revcomp          (bwtindex.c:64-172, bwt.c:385)
//   .sa   every 32nd suffix-array value                                      (bwt.c:62-84, 396-407)
// The suffix array is built by a parallel sort on 32-base packed keys with deeper comparison on ties
// (memory ~ 20 bytes per text symbol: hg38-scale text, 6.2 G symbols, needs ~130 GB of RAM and a many-core host).
//
//   mkindex <in.fa> <prefix>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <ctype.h>
#include <string>
#include <vector>
#include <algorithm>
#include <parallel/algorithm>

static const uint8_t nt4(unsigned char c)
{
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

struct Contig { std::string name, anno; int64_t offset; int32_t len, n_ambs; };
struct Hole { int64_t offset; int32_t len; char amb; };

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: mkindex <in.fa> <prefix>\n"); return 1; }
	const std::string prefix = argv[2];
	FILE *fp = fopen(argv[1], "r");
	if (!fp) { perror(argv[1]); return 1; }

	// ---- pack the FASTA (bns_fasta2bntseq / add1) ----
	std::vector<Contig> ctg;
	std::vector<Hole> holes;
	std::vector<uint8_t> base;                       // one base per byte, forward strand
	srand48(11);
	{
		std::vector<char> line(1 << 20);
		int lasts = 0;
		while (fgets(line.data(), (int)line.size(), fp)) {
			char *s = line.data();
			if (s[0] == '>') {
				char *p = s + 1, *q = p;
				while (*q && !isspace((unsigned char)*q)) ++q;
				Contig c;
				c.name.assign(p, q - p);
				while (*q && isspace((unsigned char)*q) && *q != '\n') ++q;
				char *e = q + strlen(q);
				while (e > q && (e[-1] == '\n' || e[-1] == '\r')) --e;
				c.anno = e > q ? std::string(q, e - q) : std::string("(null)");
				c.offset = (int64_t)base.size(); c.len = 0; c.n_ambs = 0;
				ctg.push_back(c);
				lasts = 0;
				continue;
			}
			if (ctg.empty()) continue;
			for (char *p = s; *p; ++p) {
				if (!isgraph((unsigned char)*p)) continue;
				int c = nt4((unsigned char)*p);
				if (c >= 4) {
					if (lasts == *p) ++holes.back().len;
					else { holes.push_back({ (int64_t)base.size(), 1, *p }); ++ctg.back().n_ambs; }
					c = (int)(lrand48() & 3);
				}
				lasts = *p;
				base.push_back((uint8_t)c);
				++ctg.back().len;
			}
		}
		fclose(fp);
	}
	const int64_t l_pac = (int64_t)base.size();
	{   // .pac (forward only)
		std::vector<uint8_t> pac((size_t)(l_pac >> 2) + ((l_pac & 3) ? 1 : 0), 0);
		for (int64_t i = 0; i < l_pac; ++i) pac[i >> 2] |= base[i] << ((~i & 3) << 1);
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

	// ---- text = forward + reverse complement ----
	const int64_t N = l_pac * 2;
	base.resize((size_t)N);
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < l_pac; ++i) base[(size_t)(N - 1 - i)] = 3 - base[(size_t)i];
	std::vector<uint64_t> w((size_t)(N + 31) / 32 + 2, 0);            // 32 bases per word, first base on top
#pragma omp parallel for schedule(static)
	for (int64_t wi = 0; wi < (N + 31) / 32; ++wi) {
		uint64_t x = 0;
		const int64_t lo = wi * 32, hi = lo + 32 < N ? lo + 32 : N;
		for (int64_t i = lo; i < hi; ++i) x |= (uint64_t)base[(size_t)i] << ((~i & 31) << 1);
		w[(size_t)wi] = x;
	}
	auto key_at = [&](int64_t i) -> uint64_t {                          // 32 bases from i, zero padded past N
		int sh = (int)(i & 31) << 1;
		uint64_t a = w[(size_t)(i >> 5)];
		return sh ? (a << sh) | (w[(size_t)(i >> 5) + 1] >> (64 - sh)) : a;
	};
	struct Ent { uint64_t key; uint64_t idx; };                         // 16 bytes either way; 64-bit idx admits hg38-scale text (6.2 G symbols)
	std::vector<Ent> sa((size_t)N);
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < N; ++i) {
		uint64_t k = key_at(i);
		int64_t rem = N - i;
		if (rem < 32) k &= ~0ull << ((32 - rem) << 1);
		sa[(size_t)i] = { k, (uint64_t)i };
	}
	auto less = [&](const Ent &a, const Ent &b) -> bool {
		if (a.key != b.key) return a.key < b.key;
		if (a.idx == b.idx) return false;
		int64_t i = a.idx, j = b.idx;
		for (int64_t d = 0;; d += 32) {
			int64_t ra = N - i - d, rb = N - j - d;
			if (ra < 32 || rb < 32) {                                    // one of them ends inside this word
				uint64_t ka = ra > 0 ? key_at(i + d) : 0, kb = rb > 0 ? key_at(j + d) : 0;
				if (ra < 32 && ra > 0) ka &= ~0ull << ((32 - ra) << 1);
				if (rb < 32 && rb > 0) kb &= ~0ull << ((32 - rb) << 1);
				if (ka != kb) return ka < kb;
				return ra < rb;                                          // the suffix that hits '$' first is smaller
			}
			uint64_t ka = key_at(i + d), kb = key_at(j + d);
			if (ka != kb) return ka < kb;
		}
	};
	__gnu_parallel::sort(sa.begin(), sa.end(), less);

	// ---- BWT, Occ interleave, sampled SA ----
	// full suffix array rows: row 0 = "$" (position N), row r+1 = sa[r]
	uint64_t primary = 0, L2[5] = { 0, 0, 0, 0, 0 };
	{
		uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma omp parallel for schedule(static) reduction(+:c0,c1,c2,c3)
		for (int64_t i = 0; i < N; ++i) { uint8_t b = base[(size_t)i]; c0 += b == 0; c1 += b == 1; c2 += b == 2; c3 += b == 3; }
		L2[1] = c0; L2[2] = c0 + c1; L2[3] = c0 + c1 + c2; L2[4] = c0 + c1 + c2 + c3;
	}
	std::vector<uint8_t> bw((size_t)N);                                  // '$'-removed BWT
	{
		int64_t rp = -1;                                                 // the row whose BWT char is '$'
#pragma omp parallel for schedule(static)
		for (int64_t r = 0; r < N; ++r) if (sa[(size_t)r].idx == 0) rp = r;
		primary = (uint64_t)rp + 1;
		bw[0] = base[(size_t)N - 1];                                     // row 0: char before '$'
#pragma omp parallel for schedule(static)
		for (int64_t r = 0; r < N; ++r) {
			if (r == rp) continue;
			const uint64_t p = sa[(size_t)r].idx;
			bw[(size_t)(r < rp ? r + 1 : r)] = base[p - 1];
		}
	}
	{
		const uint64_t n_blk = (uint64_t)(N + 127) / 128, n_occ = n_blk + 1;
		const uint64_t bwt_words = (uint64_t)(N + 15) / 16 + n_occ * 8;
		std::vector<uint32_t> buf((size_t)bwt_words, 0);
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
			uint64_t k = (uint64_t)b * 16 - 0;
			// words before block b: b*(8+8) when all earlier blocks are full (only the last block can be short)
			memcpy(&buf[(size_t)k], &cnt[(size_t)b * 4], 32);
			for (int64_t i = lo; i < hi; ++i) buf[(size_t)(k + 8 + ((i - lo) >> 4))] |= (uint32_t)bw[(size_t)i] << ((~i & 15) << 1);
		}
		{   // the trailing count block (bwtindex.c:169)
			const uint64_t k = (uint64_t)(N + 15) / 16 + n_blk * 8;
			memcpy(&buf[(size_t)k], &cnt[(size_t)n_blk * 4], 32);
		}
		FILE *f = fopen((prefix + ".bwt").c_str(), "wb");
		fwrite(&primary, 8, 1, f); fwrite(L2 + 1, 8, 4, f);
		fwrite(buf.data(), 4, buf.size(), f);
		fclose(f);
	}
	{
		const uint64_t intv = 32, n_sa = ((uint64_t)N + intv) / intv, seq_len = (uint64_t)N;
		std::vector<uint64_t> s((size_t)n_sa, 0);
		for (uint64_t j = 1; j < n_sa; ++j) s[(size_t)j] = sa[(size_t)(j * intv - 1)].idx;   // full row j*32 = sa[j*32-1]
		FILE *f = fopen((prefix + ".sa").c_str(), "wb");
		fwrite(&primary, 8, 1, f); fwrite(L2 + 1, 8, 4, f);
		fwrite(&intv, 8, 1, f); fwrite(&seq_len, 8, 1, f);
		fwrite(s.data() + 1, 8, (size_t)n_sa - 1, f);
		fclose(f);
	}
	return 0;
}
