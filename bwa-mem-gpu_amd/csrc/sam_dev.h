// SAM text on the device (mem_aln2sam, bwamem.c:832-956, and the kstring helpers it uses).  All functions are
// wavefront-collective: every lane makes the same calls with the same (uniform) arguments; `pos` advances identically
// in all lanes.  In the sizing pass (dst == nullptr) nothing is stored, so one body serves both passes and the two can
// never disagree about a length.  Bulk fields (name, SEQ, QUAL, MD) are copied one byte per lane.
#pragma once
#include "bwahip_internal.h"

namespace samdev {

struct Emit {
	uint8_t *dst; int64_t pos; int l;
	__device__ __forceinline__ void ch(char c) { if (dst && l == 0) dst[pos] = (uint8_t)c; ++pos; }
	__device__ __forceinline__ void lit(const char *s, int n) { if (dst && l < n) dst[pos + l] = (uint8_t)s[l]; pos += n; }   // n <= 64
	__device__ __forceinline__ void bytes(const uint8_t *s, int n) { if (dst) for (int i = l; i < n; i += 64) dst[pos + i] = s[i]; pos += n; }
	// NUL-terminated string of unknown length in global memory
	__device__ __forceinline__ void cstr(const uint8_t *s)
	{
		int n = 0;
		for (;; n += 64) {
			const bool z = s[n + l] == 0;                          // reads up to 63 bytes past the NUL: buffers are padded by 64
			const unsigned long long m = __ballot(z);
			if (m) { const int k = __ffsll((long long)m) - 1; if (dst && l < k) dst[pos + n + l] = s[n + l]; n += k; break; }
			if (dst) dst[pos + n + l] = s[n + l];
		}
		pos += n;
	}
	// kputw / kputl: decimal, '-' for negatives
	__device__ __forceinline__ void num(long long v)
	{
		unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
		int nd = 1;
		for (unsigned long long t = u; t >= 10; t /= 10) ++nd;
		if (v < 0) ch('-');
		if (dst && l < nd) {                                       // lane k writes the k-th digit from the left
			unsigned long long t = u;
			for (int s = nd - 1 - l; s > 0; --s) t /= 10;
			dst[pos + l] = (uint8_t)('0' + t % 10);
		}
		pos += nd;
	}
};

// "%.3f" of a positive double exactly as printf rounds it (round-half-even on the exact binary value; bwamem.c:945)
__device__ __forceinline__ void emit_f3(Emit &e, double x)
{
	unsigned long long bits = (unsigned long long)__double_as_longlong(x);
	const int ex = (int)(bits >> 52 & 0x7ff);
	unsigned long long m = bits & 0xfffffffffffffull;
	long long N;                                                   // round(x * 1000)
	if (ex == 0) N = 0;                                            // zero / subnormal
	else {
		m |= 1ull << 52;
		const int sh = 1075 - ex;                                  // x = m * 2^-sh
		const unsigned long long p = m * 1000ull;                  // < 2^63
		if (sh <= 0) N = (long long)(p << -sh);                    // (huge values are not expected here)
		else if (sh >= 64) N = 0;
		else {
			const unsigned long long q = p >> sh, rem = p & ((1ull << sh) - 1), half = 1ull << (sh - 1);
			N = (long long)(q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0));
		}
	}
	e.num(N / 1000);
	e.ch('.');
	const int fr = (int)(N % 1000);
	e.ch((char)('0' + fr / 100)); e.ch((char)('0' + fr / 10 % 10)); e.ch((char)('0' + fr % 10));
}

struct Tables {
	const uint8_t *ctg_names; const int *ctg_name_off; const uint8_t *ctg_anno; const int *ctg_anno_off;
	const uint8_t *pool; const uint8_t *rg_id; int rg_len; int opt_flag;
};
struct ReadText { const uint8_t *name, *comment, *seq /* codes 0..4 */, *qual; int l_seq; };

__device__ __forceinline__ void emit_ctg(Emit &e, const Tables &t, int rid) { e.bytes(t.ctg_names + t.ctg_name_off[rid], t.ctg_name_off[rid + 1] - t.ctg_name_off[rid]); }
__device__ __forceinline__ const uint32_t *cigar_of(const Tables &t, const DevAln &p) { return reinterpret_cast<const uint32_t*>(t.pool + p.cigar_off); }

// add_cigar (bwamem.c:819-830): clip operations print as H for a supplementary line (which != 0) unless -Y or an ALT hit
__device__ __forceinline__ void emit_cigar(Emit &e, const Tables &t, const DevAln &p, int which)
{
	if (p.n_cigar == 0) { e.ch('*'); return; }
	const uint32_t *cg = cigar_of(t, p);
	for (int i = 0; i < p.n_cigar; ++i) {
		int c = cg[i] & 0xf;
		if (!(t.opt_flag & BWAHIP_F_SOFTCLIP) && !p.is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
		e.num(cg[i] >> 4); e.ch("MIDSH"[c]);
	}
}
__device__ __forceinline__ int get_rlen(const Tables &t, const DevAln &p)   // bwamem.c:808
{
	int rl = 0;
	const uint32_t *cg = cigar_of(t, p);
	for (int i = 0; i < p.n_cigar; ++i) { const int op = cg[i] & 0xf; if (op == 0 || op == 2) rl += (int)(cg[i] >> 4); }
	return rl;
}

// One XA / XB entry (mem_gen_alt, bwamem_extra.c:147-160)
__device__ __forceinline__ void emit_xa_entry(Emit &e, const Tables &t, const DevAln &q)
{
	emit_ctg(e, t, q.rid);
	e.ch(','); e.ch("+-"[q.is_rev]); e.num(q.pos + 1); e.ch(',');
	const uint32_t *cg = cigar_of(t, q);
	for (int i = 0; i < q.n_cigar; ++i) { e.num(cg[i] >> 4); e.ch("MIDSHN"[cg[i] & 0xf]); }
	e.ch(','); e.num(q.NM);
	if (t.opt_flag & BWAHIP_F_XB) { e.ch(','); e.num(q.score); }
	e.ch(';');
}

// mem_aln2sam (bwamem.c:832-956).  list: the read's records (n of them, pointers into the alignment array), `which` the one
// to print; m: the mate's primary record or nullptr; xa_* : the XA members of this record (alignments, in order).
__device__ void emit_record(Emit &e, const Tables &t, const ReadText &s, int n, const DevAln *const *list, int which, const DevAln *m_,
                            int n_xa, const DevAln *const *xa)
{
	DevAln p = *list[which], mt;
	const bool has_m = m_ != nullptr;
	if (has_m) mt = *m_;
	p.flag |= has_m ? 0x1 : 0;
	p.flag |= p.rid < 0 ? 0x4 : 0;
	p.flag |= has_m && mt.rid < 0 ? 0x8 : 0;
	if (p.rid < 0 && has_m && mt.rid >= 0) { p.rid = mt.rid; p.pos = mt.pos; p.is_rev = mt.is_rev; p.n_cigar = 0; }   // bwamem.c:842-845
	if (has_m && mt.rid < 0 && p.rid >= 0) { mt.rid = p.rid; mt.pos = p.pos; mt.is_rev = p.is_rev; mt.n_cigar = 0; }
	p.flag |= p.is_rev ? 0x10 : 0;
	p.flag |= has_m && mt.is_rev ? 0x20 : 0;
	e.cstr(s.name); e.ch('\t');
	e.num((p.flag & 0xffff) | ((p.flag & 0x10000) ? 0x100 : 0)); e.ch('\t');
	if (p.rid >= 0) {
		emit_ctg(e, t, p.rid); e.ch('\t');
		e.num(p.pos + 1); e.ch('\t');
		e.num(p.mapq); e.ch('\t');
		emit_cigar(e, t, p, which);
	} else e.lit("*\t0\t0\t*", 7);
	e.ch('\t');
	if (has_m && mt.rid >= 0) {                                   // mate position and template length (bwamem.c:863-875)
		if (p.rid == mt.rid) e.ch('='); else emit_ctg(e, t, mt.rid);
		e.ch('\t');
		e.num(mt.pos + 1); e.ch('\t');
		if (p.rid == mt.rid) {
			const int64_t p0 = p.pos + (p.is_rev ? get_rlen(t, p) - 1 : 0);
			const int64_t p1 = mt.pos + (mt.is_rev ? get_rlen(t, mt) - 1 : 0);
			if (mt.n_cigar == 0 || p.n_cigar == 0) e.ch('0');
			else e.num(-(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else e.ch('0');
	} else e.lit("*\t0\t0", 5);
	e.ch('\t');
	if (p.flag & 0x100) e.lit("*\t*", 3);                          // secondary: no SEQ / QUAL
	else {
		int qb = 0, qe = s.l_seq;
		const bool hard = p.n_cigar && which && !(t.opt_flag & BWAHIP_F_SOFTCLIP) && !p.is_alt;
		if (hard) {
			const uint32_t *cg = cigar_of(t, p);
			const int o0 = cg[0] & 0xf, o1 = cg[p.n_cigar - 1] & 0xf;
			if (!p.is_rev) { if (o0 == 4 || o0 == 3) qb += (int)(cg[0] >> 4); if (o1 == 4 || o1 == 3) qe -= (int)(cg[p.n_cigar - 1] >> 4); }
			else { if (o0 == 4 || o0 == 3) qe -= (int)(cg[0] >> 4); if (o1 == 4 || o1 == 3) qb += (int)(cg[p.n_cigar - 1] >> 4); }
		}
		const int len = qe - qb;
		if (e.dst) {
			if (!p.is_rev) for (int i = e.l; i < len; i += 64) e.dst[e.pos + i] = (uint8_t)"ACGTN"[s.seq[qb + i]];
			else for (int i = e.l; i < len; i += 64) e.dst[e.pos + i] = (uint8_t)"TGCAN"[s.seq[qe - 1 - i]];
		}
		e.pos += len;
		e.ch('\t');
		if (s.qual) {
			if (e.dst) {
				if (!p.is_rev) for (int i = e.l; i < len; i += 64) e.dst[e.pos + i] = s.qual[qb + i];
				else for (int i = e.l; i < len; i += 64) e.dst[e.pos + i] = s.qual[qe - 1 - i];
			}
			e.pos += len;
		} else e.ch('*');
	}
	// optional tags
	if (p.n_cigar) {
		e.lit("\tNM:i:", 6); e.num(p.NM);
		e.lit("\tMD:Z:", 6); e.bytes(t.pool + p.md_off, p.md_len);
	}
	if (has_m && mt.n_cigar) { e.lit("\tMC:Z:", 6); emit_cigar(e, t, mt, which); }
	if (p.score >= 0) { e.lit("\tAS:i:", 6); e.num(p.score); }
	if (p.sub >= 0) { e.lit("\tXS:i:", 6); e.num(p.sub); }
	if (t.rg_len) { e.lit("\tRG:Z:", 6); e.bytes(t.rg_id, t.rg_len); }
	if (!(p.flag & 0x100)) {                                      // other primary hits: SA (bwamem.c:922-943)
		int i;
		for (i = 0; i < n; ++i) if (i != which && !(list[i]->flag & 0x100)) break;
		if (i < n) {
			e.lit("\tSA:Z:", 6);
			for (i = 0; i < n; ++i) {
				const DevAln &q = *list[i];
				if (i == which || (q.flag & 0x100)) continue;
				emit_ctg(e, t, q.rid); e.ch(',');
				e.num(q.pos + 1); e.ch(',');
				e.ch("+-"[q.is_rev]); e.ch(',');
				const uint32_t *cg = cigar_of(t, q);
				for (int k = 0; k < q.n_cigar; ++k) { e.num(cg[k] >> 4); e.ch("MIDSH"[cg[k] & 0xf]); }
				e.ch(','); e.num(q.mapq);
				e.ch(','); e.num(q.NM);
				e.ch(';');
			}
		}
		if (p.alt_sc > 0) { e.lit("\tpa:f:", 6); emit_f3(e, (double)p.score / p.alt_sc); }
	}
	if (n_xa > 0) {
		e.lit((t.opt_flag & BWAHIP_F_XB) ? "\tXB:Z:" : "\tXA:Z:", 6);
		for (int i = 0; i < n_xa; ++i) emit_xa_entry(e, t, *xa[i]);
	}
	if (s.comment) { e.ch('\t'); e.cstr(s.comment); }
	if ((t.opt_flag & BWAHIP_F_REF_HDR) && p.rid >= 0 && t.ctg_anno_off[p.rid + 1] > t.ctg_anno_off[p.rid]) {   // XR (bwamem.c:948-954): tabs become spaces
		e.lit("\tXR:Z:", 6);
		const uint8_t *an = t.ctg_anno + t.ctg_anno_off[p.rid];
		const int la = t.ctg_anno_off[p.rid + 1] - t.ctg_anno_off[p.rid];
		if (e.dst) for (int i = e.l; i < la; i += 64) e.dst[e.pos + i] = an[i] == '\t' ? (uint8_t)' ' : an[i];
		e.pos += la;
	}
	e.ch('\n');
}

} // namespace samdev
