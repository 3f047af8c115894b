// TEST INFRASTRUCTURE, not part of libbwahip.so: a second, independent implementation of everything mem_process_seqs does AFTER the hot
// path (worker2, bwamem.c:1197) -- mark primary / mapQ / CIGAR+NM+MD / SAM text for single-end reads, and for pairs insert-size
// statistics, mate rescue, pairing and paired SAM output -- on host threads, on top of the library's public C ABI (bwahip_align_batch
// gives it the regions the GPU hot path found).  It is built as libbwahip_hostfinal.so; the product's finalisation are the GPU kernels of
// k_final.hip / k_pair.hip / k_sam.hip, which the tests cross-check against this one (knobs gpu_final = 0 / gpu_pair = 0 make
// bwahip_process_seqs load it).  Round 1's product path; kept because two implementations that agree byte for byte pin each other.
//
// Reference semantics restated here (file:line in /root/reference):
//   bwamem.c:500-565 mark primary      bwamem.c:962-986 mapQ          bwamem.c:988-1010 primary5 reorder
//   bwamem.c:1099-1170 mem_reg2aln      bwa.c:261-347 CIGAR/NM/MD      ksw.c:504-606 banded global alignment
//   bwamem.c:799-956 SAM record         bwamem.c:1013-1059 mem_reg2sam  bwamem_extra.c:116-169 XA tag
//   bwamem_pair.c:48-419 PE logic       ksw.c:64-365 SSE2 striped local alignment (emulated lane by lane)
//   ksort.h:146-227 unstable introsort (permutation of ties is result-visible), utils.h:97 hash_64
#include "bwahip_internal.h"
#include <math.h>
#include <limits.h>
#include <atomic>
#include <thread>
#include <algorithm>

namespace hf {

typedef bwahip_alnreg_t Reg;
typedef bwahip_opt_t Opt;
struct RegV { std::vector<Reg> a; };

struct Aln {                       // mem_aln_t, bwa.h:173-184
	int64_t pos = 0;
	std::string XA; bool has_XA = false;
	std::vector<uint32_t> cigar; std::string md;
	int rid = 0, flag = 0;
	uint32_t is_rev = 0, is_alt = 0, mapq = 0, NM = 0;
	int score = 0, sub = 0, alt_sc = 0;
	int n_cigar() const { return (int)cigar.size(); }
};

struct Ref { const bwahip_bns_t *bns; const uint8_t *pac; int64_t l_pac; const char *rg_id; };   // rg_id: bwa_rg_id (bwa.c:44), "" = none

// ---------------------------------------------------------------- ksort.h:146-227
template <class T, class LT> static void insertion(T *s, T *t, LT lt)
{
	for (T *i = s + 1; i < t; ++i)
		for (T *j = i; j > s && lt(*j, *(j - 1)); --j) std::swap(*j, *(j - 1));
}
template <class T, class LT> static void combsort(size_t n, T *a, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	bool swapped;
	size_t gap = n;
	do {
		if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (T *i = a; i < a + n - gap; ++i) { T *j = i + gap; if (lt(*j, *i)) { std::swap(*i, *j); swapped = true; } }
	} while (swapped || gap > 2);
	if (gap != 1) insertion(a, a + n, lt);
}
template <class T, class LT> static void introsort(size_t n, T *a, LT lt)
{
	struct Frame { T *lo, *hi; int depth; };
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) std::swap(a[0], a[1]); return; }
	int d;
	for (d = 2; 1ul << d < n; ++d);
	std::vector<Frame> stack(sizeof(size_t) * d + 2);
	Frame *top = stack.data();
	T *s = a, *t = a + (n - 1), *i, *j, *k;
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { combsort((size_t)(t - s) + 1, s, lt); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
			else k = lt(*j, *i) ? i : j;
			T pivot = *k;
			if (k != t) std::swap(*k, *t);
			for (;;) {
				do ++i; while (lt(*i, pivot));
				do --j; while (i <= j && lt(pivot, *j));
				if (j <= i) break;
				std::swap(*i, *j);
			}
			std::swap(*i, *t);
			if (i - s > t - i) {
				if (i - s > 16) { top->lo = s; top->hi = i - 1; top->depth = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { top->lo = i + 1; top->hi = t; top->depth = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == stack.data()) { insertion(a, a + n, lt); return; }
			--top; s = top->lo; t = top->hi; d = top->depth;
		}
	}
}

static inline uint64_t hash_64(uint64_t key)      // utils.h:97
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

static const uint8_t nt4[256] = {
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,5,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4
};

// ---------------------------------------------------------------- reference sequence access (bntseq.c:354-451)
static int pos2rid(const Ref &r, int64_t pos_f)
{
	if (pos_f >= r.l_pac) return -1;
	int left = 0, mid = 0, right = r.bns->n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= r.bns->anns[mid].offset) {
			if (mid == r.bns->n_seqs - 1) break;
			if (pos_f < r.bns->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
static inline int64_t depos(const Ref &r, int64_t pos, int *is_rev) { return (*is_rev = (pos >= r.l_pac)) ? (r.l_pac << 1) - 1 - pos : pos; }
static inline int pac_at(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }
static bool get_seq(const Ref &r, int64_t beg, int64_t end, std::vector<uint8_t> &seq)   // bns_get_seq; false if bridging
{
	if (end < beg) std::swap(beg, end);
	if (end > r.l_pac << 1) end = r.l_pac << 1;
	if (beg < 0) beg = 0;
	seq.clear();
	if (beg >= r.l_pac || end <= r.l_pac) {
		seq.reserve((size_t)(end - beg));
		if (beg >= r.l_pac) {
			int64_t beg_f = (r.l_pac << 1) - 1 - end, end_f = (r.l_pac << 1) - 1 - beg;
			for (int64_t k = end_f; k > beg_f; --k) seq.push_back((uint8_t)(3 - pac_at(r.pac, k)));
		} else for (int64_t k = beg; k < end; ++k) seq.push_back((uint8_t)pac_at(r.pac, k));
		return true;
	}
	return false;
}
static void fetch_seq(const Ref &r, int64_t *beg, int64_t mid, int64_t *end, int *rid, std::vector<uint8_t> &seq)   // bns_fetch_seq
{
	int is_rev;
	if (*end < *beg) std::swap(*beg, *end);
	*rid = pos2rid(r, depos(r, mid, &is_rev));
	int64_t far_beg = r.bns->anns[*rid].offset, far_end = far_beg + r.bns->anns[*rid].len;
	if (is_rev) { int64_t t = far_beg; far_beg = (r.l_pac << 1) - far_end; far_end = (r.l_pac << 1) - t; }
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	get_seq(r, *beg, *end, seq);
}

// ---------------------------------------------------------------- ksw_global2 (ksw.c:504-606)
struct EH { int32_t h, e; };
static const int NEG_INF = -0x40000000;
static inline void push_cigar(std::vector<uint32_t> &c, int op, int len)
{
	if (c.empty() || op != (int)(c.back() & 0xf)) c.push_back((uint32_t)len << 4 | op);
	else c.back() += (uint32_t)len << 4;
}
static int global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins,
                   int w, std::vector<uint32_t> *cigar)
{
	const int m = 5, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	std::vector<uint8_t> z(cigar ? (size_t)n_col * tlen + 1 : 0);
	std::vector<int8_t> qp((size_t)qlen * m);
	std::vector<EH> eh(qlen + 1);
	for (int k = 0, i = 0; k < m; ++k) for (int j = 0; j < qlen; ++j) qp[i++] = mat[k * m + query[j]];
	eh[0].h = 0; eh[0].e = NEG_INF;
	int j;
	for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(o_ins + e_ins * j); eh[j].e = NEG_INF; }
	for (; j <= qlen; ++j) eh[j].h = eh[j].e = NEG_INF;
	for (int i = 0; i < tlen; ++i) {
		int32_t f = NEG_INF, h1, beg, end, t;
		const int8_t *q = &qp[(size_t)target[i] * qlen];
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		uint8_t *zi = cigar ? &z[(size_t)i * n_col] : nullptr;
		for (j = beg; j < end; ++j) {
			EH *p = &eh[j];
			int32_t h, M = p->h, e = p->e;
			uint8_t d;
			p->h = h1;
			M += q[j];
			d = M >= e ? 0 : 1;
			h = M >= e ? M : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = M - oe_del; e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e = e > t ? e : t;
			p->e = e;
			t = M - oe_ins; f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f = f > t ? f : t;
			if (zi) zi[j - beg] = d;
		}
		eh[end].h = h1; eh[end].e = NEG_INF;
	}
	int score = eh[qlen].h;
	if (cigar) {
		int which = 0, i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		cigar->clear();
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { push_cigar(*cigar, 0, 1); --i; --k; }
			else if (which == 1) { push_cigar(*cigar, 2, 1); --i; }
			else { push_cigar(*cigar, 1, 1); --k; }
		}
		if (i >= 0) push_cigar(*cigar, 2, i + 1);
		if (k >= 0) push_cigar(*cigar, 1, k + 1);
		std::reverse(cigar->begin(), cigar->end());
	}
	return score;
}

static void put_int(std::string &s, long v) { char b[32]; int n = snprintf(b, sizeof b, "%ld", v); s.append(b, n); }

// bwa_gen_cigar2 (bwa.c:261-347).  Returns false when the reference returns NULL (no CIGAR); NM=-1 then.
static bool gen_cigar2(const Opt &o, int w_, const Ref &r, int l_query, uint8_t *query, int64_t rb, int64_t re, int *score,
                       std::vector<uint32_t> *cigar, std::string *md, int *NM)
{
	std::vector<uint8_t> rseq;
	if (cigar) cigar->clear();
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < r.l_pac && re > r.l_pac)) return false;
	get_seq(r, rb, re, rseq);
	const int64_t rlen = (int64_t)rseq.size();
	if (re - rb != rlen) return false;
	if (rb >= r.l_pac) { std::reverse(query, query + l_query); std::reverse(rseq.begin(), rseq.end()); }
	if (l_query == re - rb && w_ == 0) {
		if (cigar) cigar->assign(1, (uint32_t)l_query << 4 | 0);
		*score = 0;
		for (int i = 0; i < l_query; ++i) *score += o.mat[rseq[i] * 5 + query[i]];
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_ins) / o.e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * o.mat[0] - o.o_del) / o.e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		int w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		int min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = global2(l_query, query, (int)rlen, rseq.data(), o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, w, cigar);
	}
	if (NM && cigar) {
		int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
		const char *int2base = rb < r.l_pac ? "ACGTN" : "TGCAN";
		md->clear();
		const int n_cigar = (int)cigar->size();
		for (int k = 0; k < n_cigar; ++k) {
			int op = (*cigar)[k] & 0xf, len = (*cigar)[k] >> 4;
			if (op == 0) {
				for (int i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) { put_int(*md, u); md->push_back(int2base[rseq[y + i]]); ++n_mm; u = 0; }
					else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < n_cigar - 1) {
					put_int(*md, u); md->push_back('^');
					for (int i = 0; i < len; ++i) md->push_back(int2base[rseq[y + i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) { x += len; n_gap += len; }
		}
		put_int(*md, u);
		*NM = n_mm + n_gap;
	}
	if (rb >= r.l_pac) std::reverse(query, query + l_query);
	return true;
}

// ---------------------------------------------------------------- mark primary (bwamem.c:500-565)
static void mark_primary_core(const Opt &o, int n, Reg *a, std::vector<int> &z)
{
	int tmp = o.a + o.b;
	tmp = o.o_del + o.e_del > tmp ? o.o_del + o.e_del : tmp;
	tmp = o.o_ins + o.e_ins > tmp ? o.o_ins + o.e_ins : tmp;
	z.clear(); z.push_back(0);
	for (int i = 1; i < n; ++i) {
		size_t k;
		for (k = 0; k < z.size(); ++k) {
			int j = z[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * o.mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == z.size()) z.push_back(i);
		else a[i].secondary = z[k];
	}
}
static int mark_primary_se(const Opt &o, int n, Reg *a, int64_t id)
{
	int n_pri = 0;
	std::vector<int> z;
	if (n == 0) return 0;
	for (int i = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64((uint64_t)(id + i));
		if (!a[i].is_alt) ++n_pri;
	}
	introsort((size_t)n, a, [](const Reg &x, const Reg &y) {
		return x.score > y.score || (x.score == y.score && (x.is_alt < y.is_alt || (x.is_alt == y.is_alt && x.hash < y.hash))); });
	mark_primary_core(o, n, a, z);
	for (int i = 0; i < n; ++i) {
		Reg *p = &a[i];
		p->secondary_all = i;
		if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {
		z.assign(n, 0);
		if (n_pri > 0) introsort((size_t)n, a, [](const Reg &x, const Reg &y) {
			return x.is_alt < y.is_alt || (x.is_alt == y.is_alt && (x.score > y.score || (x.score == y.score && x.hash < y.hash))); });
		for (int i = 0; i < n; ++i) z[a[i].secondary_all] = i;
		for (int i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) { a[i].secondary_all = z[a[i].secondary]; if (a[i].is_alt) a[i].secondary = INT_MAX; }
			else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {
			for (int i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
			mark_primary_core(o, n_pri, a, z);
		}
	} else for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	return n_pri;
}

static int approx_mapq_se(const Opt &o, const Reg *a)   // bwamem.c:962
{
	int mapq, l, sub = a->sub ? a->sub : o.min_seed_len * o.a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * o.a - a->score) / (o.a + o.b) / l;
	if (a->score == 0) mapq = 0;
	else if (o.mapQ_coef_len > 0) {
		double tmp = l < o.mapQ_coef_len ? 1. : o.mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / o.a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

static void reorder_primary5(int T, RegV &a)   // bwamem.c:988
{
	int n_pri = 0, left_st = INT_MAX, left_k = -1, n = (int)a.a.size();
	for (int k = 0; k < n; ++k) if (a.a[k].secondary < 0 && !a.a[k].is_alt && a.a[k].score >= T) ++n_pri;
	if (n_pri <= 1) return;
	for (int k = 0; k < n; ++k) {
		Reg *p = &a.a[k];
		if (p->secondary >= 0 || p->is_alt || p->score < T) continue;
		if (p->qb < left_st) { left_st = p->qb; left_k = k; }
	}
	if (left_k == 0) return;
	std::swap(a.a[0], a.a[left_k]);
	for (int k = 1; k < n; ++k) {
		Reg *p = &a.a[k];
		if (p->secondary == 0) p->secondary = left_k; else if (p->secondary == left_k) p->secondary = 0;
		if (p->secondary_all == 0) p->secondary_all = left_k; else if (p->secondary_all == left_k) p->secondary_all = 0;
	}
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)   // bwamem.c:799
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = (int)(((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.));
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

static Aln reg2aln(const Opt &o, const Ref &r, int l_query, const char *query_, const Reg *ar)   // bwamem.c:1099
{
	Aln a;
	if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	int qb = ar->qb, qe = ar->qe, NM = -1, score = 0, last_sc = -(1 << 30), is_rev;
	int64_t rb = ar->rb, re = ar->re;
	std::vector<uint8_t> query(l_query);
	for (int i = 0; i < l_query; ++i) query[i] = query_[i] < 5 ? (uint8_t)query_[i] : nt4[(uint8_t)query_[i]];
	a.mapq = ar->secondary < 0 ? (uint32_t)approx_mapq_se(o, ar) & 0xff : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	int tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, o.a, o.o_del, o.e_del);
	int w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, o.a, o.o_ins, o.e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > o.w) w2 = w2 < ar->w ? w2 : ar->w;
	int i = 0;
	do {
		w2 = w2 < o.w << 2 ? w2 : o.w << 2;
		gen_cigar2(o, w2, r, qe - qb, &query[qb], rb, re, &score, &a.cigar, &a.md, &NM);
		if (score == last_sc || w2 == o.w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - o.a);
	a.NM = (uint32_t)NM & 0x3fffff;
	int64_t pos = depos(r, rb < r.l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (!a.cigar.empty()) {
		if ((a.cigar[0] & 0xf) == 2) { pos += a.cigar[0] >> 4; a.cigar.erase(a.cigar.begin()); }
		else if ((a.cigar.back() & 0xf) == 2) a.cigar.pop_back();
	}
	if (qb != 0 || qe != l_query) {
		int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) a.cigar.insert(a.cigar.begin(), (uint32_t)clip5 << 4 | 3);
		if (clip3) a.cigar.push_back((uint32_t)clip3 << 4 | 3);
	}
	a.rid = pos2rid(r, pos);
	a.pos = pos - r.bns->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	return a;
}

// ---------------------------------------------------------------- SAM record (bwamem.c:808-956)
static inline int get_rlen(const Aln &p) { int l = 0; for (uint32_t c : p.cigar) { int op = c & 0xf; if (op == 0 || op == 2) l += c >> 4; } return l; }
static void add_cigar(const Opt &o, const Aln &p, std::string &s, int which)
{
	if (p.n_cigar()) {
		for (uint32_t cg : p.cigar) {
			int c = cg & 0xf;
			if (!(o.flag & BWAHIP_F_SOFTCLIP) && !p.is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
			put_int(s, cg >> 4); s.push_back("MIDSH"[c]);
		}
	} else s.push_back('*');
}
struct Read { const char *name, *comment, *qual; const char *seq; int l_seq; };

static void aln2sam(const Opt &o, const Ref &r, std::string &str, const Read &s, int n, const Aln *list, int which, const Aln *m_)
{
	Aln p = list[which], mtmp; Aln *m = 0;
	if (m_) { mtmp = *m_; m = &mtmp; }
	p.flag |= m ? 0x1 : 0;
	p.flag |= p.rid < 0 ? 0x4 : 0;
	p.flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p.rid < 0 && m && m->rid >= 0) { p.rid = m->rid; p.pos = m->pos; p.is_rev = m->is_rev; p.cigar.clear(); }
	if (m && m->rid < 0 && p.rid >= 0) { m->rid = p.rid; m->pos = p.pos; m->is_rev = p.is_rev; m->cigar.clear(); }
	p.flag |= p.is_rev ? 0x10 : 0;
	p.flag |= m && m->is_rev ? 0x20 : 0;
	str += s.name; str.push_back('\t');
	put_int(str, (p.flag & 0xffff) | (p.flag & 0x10000 ? 0x100 : 0)); str.push_back('\t');
	if (p.rid >= 0) {
		str += r.bns->anns[p.rid].name; str.push_back('\t');
		put_int(str, p.pos + 1); str.push_back('\t');
		put_int(str, p.mapq); str.push_back('\t');
		add_cigar(o, p, str, which);
	} else str += "*\t0\t0\t*";
	str.push_back('\t');
	if (m && m->rid >= 0) {
		if (p.rid == m->rid) str.push_back('='); else str += r.bns->anns[m->rid].name;
		str.push_back('\t');
		put_int(str, m->pos + 1); str.push_back('\t');
		if (p.rid == m->rid) {
			int64_t p0 = p.pos + (p.is_rev ? get_rlen(p) - 1 : 0);
			int64_t p1 = m->pos + (m->is_rev ? get_rlen(*m) - 1 : 0);
			if (m->n_cigar() == 0 || p.n_cigar() == 0) str.push_back('0');
			else put_int(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else str.push_back('0');
	} else str += "*\t0\t0";
	str.push_back('\t');
	if (p.flag & 0x100) str += "*\t*";
	else if (!p.is_rev) {
		int qb = 0, qe = s.l_seq;
		if (p.n_cigar() && which && !(o.flag & BWAHIP_F_SOFTCLIP) && !p.is_alt) {
			if ((p.cigar[0] & 0xf) == 4 || (p.cigar[0] & 0xf) == 3) qb += p.cigar[0] >> 4;
			if ((p.cigar.back() & 0xf) == 4 || (p.cigar.back() & 0xf) == 3) qe -= p.cigar.back() >> 4;
		}
		for (int i = qb; i < qe; ++i) str.push_back("ACGTN"[(int)s.seq[i]]);
		str.push_back('\t');
		if (s.qual) str.append(s.qual + qb, qe - qb); else str.push_back('*');
	} else {
		int qb = 0, qe = s.l_seq;
		if (p.n_cigar() && which && !(o.flag & BWAHIP_F_SOFTCLIP) && !p.is_alt) {
			if ((p.cigar[0] & 0xf) == 4 || (p.cigar[0] & 0xf) == 3) qe -= p.cigar[0] >> 4;
			if ((p.cigar.back() & 0xf) == 4 || (p.cigar.back() & 0xf) == 3) qb += p.cigar.back() >> 4;
		}
		for (int i = qe - 1; i >= qb; --i) str.push_back("TGCAN"[(int)s.seq[i]]);
		str.push_back('\t');
		if (s.qual) for (int i = qe - 1; i >= qb; --i) str.push_back(s.qual[i]); else str.push_back('*');
	}
	if (p.n_cigar()) { str += "\tNM:i:"; put_int(str, p.NM); str += "\tMD:Z:"; str += p.md; }
	if (m && m->n_cigar()) { str += "\tMC:Z:"; add_cigar(o, *m, str, which); }
	if (p.score >= 0) { str += "\tAS:i:"; put_int(str, p.score); }
	if (p.sub >= 0) { str += "\tXS:i:"; put_int(str, p.sub); }
	if (r.rg_id && r.rg_id[0]) { str += "\tRG:Z:"; str += r.rg_id; }
	if (!(p.flag & 0x100)) {
		int i;
		for (i = 0; i < n; ++i) if (i != which && !(list[i].flag & 0x100)) break;
		if (i < n) {
			str += "\tSA:Z:";
			for (i = 0; i < n; ++i) {
				const Aln &q = list[i];
				if (i == which || (q.flag & 0x100)) continue;
				str += r.bns->anns[q.rid].name; str.push_back(',');
				put_int(str, q.pos + 1); str.push_back(',');
				str.push_back("+-"[q.is_rev]); str.push_back(',');
				for (uint32_t cg : q.cigar) { put_int(str, cg >> 4); str.push_back("MIDSH"[cg & 0xf]); }
				str.push_back(','); put_int(str, q.mapq);
				str.push_back(','); put_int(str, q.NM);
				str.push_back(';');
			}
		}
		if (p.alt_sc > 0) { char b[64]; snprintf(b, sizeof b, "\tpa:f:%.3f", (double)p.score / p.alt_sc); str += b; }
	}
	if (p.has_XA) { str += (o.flag & BWAHIP_F_XB) ? "\tXB:Z:" : "\tXA:Z:"; str += p.XA; }
	if (s.comment) { str.push_back('\t'); str += s.comment; }
	if ((o.flag & BWAHIP_F_REF_HDR) && p.rid >= 0 && r.bns->anns[p.rid].anno != 0 && r.bns->anns[p.rid].anno[0] != 0) {
		str += "\tXR:Z:";
		size_t t0 = str.size();
		str += r.bns->anns[p.rid].anno;
		for (; t0 < str.size(); ++t0) if (str[t0] == '\t') str[t0] = ' ';
	}
	str.push_back('\n');
}

// mem_gen_alt (bwamem_extra.c:116-169): per region index, the XA string (has[k] false = NULL in the reference)
static void gen_alt(const Opt &o, const Ref &r, const RegV &a, int l_query, const char *query, std::vector<std::string> &XA, std::vector<char> &has, bool &any)
{
	const int n = (int)a.a.size();
	std::vector<int> cnt(n, 0);
	std::vector<char> has_alt(n, 0);
	int tot = 0;
	auto pri = [&](int i) { int k = a.a[i].secondary_all; return (k >= 0 && a.a[i].score >= a.a[k].score * (double)o.XA_drop_ratio) ? k : -1; };
	for (int i = 0; i < n; ++i) { int k = pri(i); if (k >= 0) { ++cnt[k]; ++tot; if (a.a[i].is_alt) has_alt[k] = 1; } }
	XA.assign(n, std::string()); has.assign(n, 0);
	any = tot != 0;
	if (!any) return;
	for (int i = 0; i < n; ++i) {
		int k = pri(i);
		if (k < 0) continue;
		if (cnt[k] > o.max_XA_hits_alt || (!has_alt[k] && cnt[k] > o.max_XA_hits)) continue;
		Aln t = reg2aln(o, r, l_query, query, &a.a[i]);
		std::string &s = XA[k];
		s += r.bns->anns[t.rid].name;
		s.push_back(','); s.push_back("+-"[t.is_rev]); put_int(s, t.pos + 1);
		s.push_back(',');
		for (uint32_t cg : t.cigar) { put_int(s, cg >> 4); s.push_back("MIDSHN"[cg & 0xf]); }
		s.push_back(','); put_int(s, t.NM);
		if (o.flag & BWAHIP_F_XB) { s.push_back(','); put_int(s, t.score); }
		s.push_back(';');
		has[k] = 1;
	}
}

static char *dup_sam(const std::string &s) { char *p = (char*)malloc(s.size() + 1); if (p) { memcpy(p, s.data(), s.size()); p[s.size()] = 0; } return p; }

static void reg2sam(const Opt &o, const Ref &r, const Read &s, RegV &a, int extra_flag, const Aln *m, char **sam)   // bwamem.c:1013
{
	std::string str;
	std::vector<Aln> aa;
	std::vector<std::string> XA; std::vector<char> has; bool any = false;
	const bool want_XA = !(o.flag & BWAHIP_F_ALL);
	if (want_XA) gen_alt(o, r, a, s.l_seq, s.seq, XA, has, any);
	int l = 0;
	for (size_t k = 0; k < a.a.size(); ++k) {
		Reg *p = &a.a[k];
		if (p->score < o.T) continue;
		if (p->secondary >= 0 && (p->is_alt || !(o.flag & BWAHIP_F_ALL))) continue;
		if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < a.a[p->secondary].score * o.drop_ratio) continue;
		aa.push_back(reg2aln(o, r, s.l_seq, s.seq, p));
		Aln &q = aa.back();
		if (want_XA && any && has[k]) { q.XA = XA[k]; q.has_XA = true; }
		q.flag |= extra_flag;
		if (p->secondary >= 0) q.sub = -1;
		if (l && p->secondary < 0) q.flag |= (o.flag & BWAHIP_F_NO_MULTI) ? 0x10000 : 0x800;
		if (!(o.flag & BWAHIP_F_KEEP_SUPP_MAPQ) && l && !p->is_alt && q.mapq > aa[0].mapq) q.mapq = aa[0].mapq;
		++l;
	}
	if (aa.empty()) {
		Aln t = reg2aln(o, r, s.l_seq, s.seq, 0);
		t.flag |= extra_flag;
		aln2sam(o, r, str, s, 1, &t, 0, m);
	} else for (size_t k = 0; k < aa.size(); ++k) aln2sam(o, r, str, s, (int)aa.size(), aa.data(), (int)k, m);
	*sam = dup_sam(str);
}

// ---------------------------------------------------------------- striped local alignment (ksw.c:64-365), emulated lane by lane
struct Kswr { int score, te, qe, score2, te2, tb, qb; };
enum { XBYTE = 0x10000, XSTOP = 0x20000, XSUBO = 0x40000, XSTART = 0x80000 };
struct SProf { int qlen, slen, size, P, shift, mdiff, max; std::vector<int> prof, H0, H1, E, Hmax; };
static void sprof_init(SProf &q, int size, int qlen, const uint8_t *query, const int8_t *mat)
{
	const int m = 5;
	int mn = 127, mx = 0;
	size = size > 1 ? 2 : 1;
	q.size = size; q.P = 8 * (3 - size); q.qlen = qlen; q.slen = (qlen + q.P - 1) / q.P;
	for (int a = 0; a < m * m; ++a) { if (mat[a] < (int8_t)mn) mn = (uint8_t)mat[a]; if (mat[a] > (int8_t)mx) mx = (uint8_t)mat[a]; }
	q.max = mx; q.shift = (256 - mn) & 0xff; q.mdiff = (mx + q.shift) & 0xff;
	const int sz = q.slen * q.P;
	q.prof.assign((size_t)m * sz, 0); q.H0.assign(sz, 0); q.H1.assign(sz, 0); q.E.assign(sz, 0); q.Hmax.assign(sz, 0);
	for (int a = 0; a < m; ++a)
		for (int i = 0; i < q.slen; ++i)
			for (int l = 0, k = i; l < q.P; ++l, k += q.slen) {
				int v = k >= qlen ? 0 : mat[a * m + query[k]];
				q.prof[(size_t)(a * q.slen + i) * q.P + l] = size == 1 ? ((v + q.shift) & 0xff) : v;
			}
}
static inline int sat_add_u8(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
static inline int sat_sub_u(int a, int b) { int s = a - b; return s < 0 ? 0 : s; }
static inline int sat_add_i16(int a, int b) { int s = a + b; return s > 32767 ? 32767 : s < -32768 ? -32768 : s; }
static Kswr striped_sw(SProf &q, int tlen, const uint8_t *target, int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	const int P = q.P, slen = q.slen; const bool is8 = q.size == 1;
	int te = -1, gmax = 0, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int minsc = (xtra & XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & XSTOP) ? xtra & 0xffff : 0x10000;
	std::vector<uint64_t> b;
	int *H0 = q.H0.data(), *H1 = q.H1.data(), *E = q.E.data(), *Hmax = q.Hmax.data();
	std::vector<int> hv(P), fv(P), mxv(P);
	Kswr r = { 0, -1, -1, -1, -1, -1, -1 };
	if (is8) { oe_del &= 0xff; oe_ins &= 0xff; e_del &= 0xff; e_ins &= 0xff; }
	std::fill(q.E.begin(), q.E.end(), 0); std::fill(q.H0.begin(), q.H0.end(), 0); std::fill(q.Hmax.begin(), q.Hmax.end(), 0);
	for (int i = 0; i < tlen; ++i) {
		const int *S = q.prof.data() + (size_t)target[i] * slen * P;
		for (int l = 0; l < P; ++l) { fv[l] = 0; mxv[l] = 0; }
		hv[0] = 0;
		for (int l = 1; l < P; ++l) hv[l] = H0[(slen - 1) * P + l - 1];
		for (int j = 0; j < slen; ++j)
			for (int l = 0; l < P; ++l) {
				int h, e = E[j * P + l], t;
				if (is8) { h = sat_add_u8(hv[l], S[j * P + l]); h = sat_sub_u(h, q.shift); }
				else h = sat_add_i16(hv[l], S[j * P + l]);
				h = h > e ? h : e;
				h = h > fv[l] ? h : fv[l];
				mxv[l] = mxv[l] > h ? mxv[l] : h;
				H1[j * P + l] = h;
				e = sat_sub_u(e, e_del); t = sat_sub_u(h, oe_del);
				E[j * P + l] = e > t ? e : t;
				fv[l] = sat_sub_u(fv[l], e_ins); t = sat_sub_u(h, oe_ins);
				fv[l] = fv[l] > t ? fv[l] : t;
				hv[l] = H0[j * P + l];
			}
		bool done = false;
		for (int k = 0; k < 16 && !done; ++k) {
			for (int l = P - 1; l > 0; --l) fv[l] = fv[l - 1];
			fv[0] = 0;
			for (int j = 0; j < slen; ++j) {
				bool any = false;
				for (int l = 0; l < P; ++l) {
					int h = H1[j * P + l];
					h = h > fv[l] ? h : fv[l];
					H1[j * P + l] = h;
					h = sat_sub_u(h, oe_ins);
					fv[l] = sat_sub_u(fv[l], e_ins);
					if (fv[l] > h) any = true;
				}
				if (!any) { done = true; break; }
			}
		}
		int imax = 0;
		for (int l = 0; l < P; ++l) imax = imax > mxv[l] ? imax : mxv[l];
		if (imax >= minsc) {
			if (b.empty() || (int32_t)b.back() + 1 != i) b.push_back((uint64_t)imax << 32 | (uint32_t)i);
			else if ((int)(b.back() >> 32) < imax) b.back() = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, sizeof(int) * slen * P);
			if (is8 ? (gmax + q.shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;
		}
		std::swap(H0, H1);
	}
	r.score = is8 ? (gmax + q.shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!is8 || r.score != 255) {
		int max = -1, tmp, n = slen * P;
		if (!is8) r.qe = -1;
		for (int i = 0; i < n; ++i) {
			int v = Hmax[i];
			if (v > max) { max = v; r.qe = i / P + i % P * slen; }
			else if (v == max && (tmp = i / P + i % P * slen) < r.qe) r.qe = tmp;
		}
		if (!b.empty()) {
			int i = (r.score + q.max - 1) / q.max, low = te - i, high = te + i;
			for (uint64_t x : b) { int e2 = (int32_t)x; if ((e2 < low || e2 > high) && (int)(x >> 32) > r.score2) { r.score2 = (int)(x >> 32); r.te2 = e2; } }
		}
	}
	return r;
}
static Kswr ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	SProf q;
	sprof_init(q, (xtra & XBYTE) ? 1 : 2, qlen, query, mat);
	const int size = q.size;
	Kswr r = striped_sw(q, tlen, target, o_del, e_del, o_ins, e_ins, xtra), rr;
	if ((xtra & XSTART) == 0 || ((xtra & XSUBO) && r.score < (xtra & 0xffff))) return r;
	std::reverse(query, query + r.qe + 1); std::reverse(target, target + r.te + 1);
	SProf q2;
	sprof_init(q2, size, r.qe + 1, query, mat);
	rr = striped_sw(q2, tlen, target, o_del, e_del, o_ins, e_ins, XSTOP | r.score);   // ksw.c:359 scans tlen, not te+1
	std::reverse(query, query + r.qe + 1); std::reverse(target, target + r.te + 1);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

// ---------------------------------------------------------------- sort/dedup without patching (mem_sort_dedup_patch with bns == 0, as mem_matesw calls it)
static int sort_dedup_nopatch(const Opt &o, int n, Reg *a)   // bwamem.c:444-496, mem_patch_reg returns 0 when bns == 0 (bwamem.c:417)
{
	if (n <= 1) return n;
	introsort((size_t)n, a, [](const Reg &x, const Reg &y) { return x.re < y.re; });
	for (int i = 0; i < n; ++i) a[i].n_comp = 1;
	for (int i = 1; i < n; ++i) {
		Reg *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + o.max_chain_gap) continue;
		for (int j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + o.max_chain_gap; --j) {
			Reg *q = &a[j];
			if (q->qe == q->qb) continue;
			int64_t orr = q->re - p->rb, oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (orr > o.mask_level_redun * mr && oq > o.mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			}
		}
	}
	int m = 0;
	for (int i = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	introsort((size_t)n, a, [](const Reg &x, const Reg &y) {
		return x.score > y.score || (x.score == y.score && (x.rb < y.rb || (x.rb == y.rb && x.qb < y.qb))); });
	for (int i = 1; i < n; ++i) if (a[i].score == a[i-1].score && a[i].rb == a[i-1].rb && a[i].qb == a[i-1].qb) a[i].qe = a[i].qb;
	m = n > 0 ? 1 : 0;
	for (int i = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}

// ---------------------------------------------------------------- PE (bwamem_pair.c)
typedef bwahip_pestat_t PeStat;
static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}
static int cal_sub(const Opt &o, const RegV &r)
{
	size_t j;
	for (j = 1; j < r.a.size(); ++j) {
		int b_max = r.a[j].qb > r.a[0].qb ? r.a[j].qb : r.a[0].qb;
		int e_min = r.a[j].qe < r.a[0].qe ? r.a[j].qe : r.a[0].qe;
		if (e_min > b_max) {
			int min_l = r.a[j].qe - r.a[j].qb < r.a[0].qe - r.a[0].qb ? r.a[j].qe - r.a[j].qb : r.a[0].qe - r.a[0].qb;
			if (e_min - b_max >= min_l * o.mask_level) break;
		}
	}
	return j < r.a.size() ? r.a[j].score : o.min_seed_len * o.a;
}
static void pestat(const Opt &o, int64_t l_pac, int n, const RegV *regs, PeStat pes[4], int verbose)   // bwamem_pair.c:72
{
	std::vector<uint64_t> isize[4];
	memset(pes, 0, 4 * sizeof(PeStat));
	for (int i = 0; i < n >> 1; ++i) {
		const RegV &r0 = regs[i << 1 | 0], &r1 = regs[i << 1 | 1];
		int64_t is;
		if (r0.a.empty() || r1.a.empty()) continue;
		if (cal_sub(o, r0) > 0.8 * r0.a[0].score) continue;
		if (cal_sub(o, r1) > 0.8 * r1.a[0].score) continue;
		if (r0.a[0].rid != r1.a[0].rid) continue;
		int dir = infer_dir(l_pac, r0.a[0].rb, r1.a[0].rb, &is);
		if (is && is <= o.max_ins) isize[dir].push_back((uint64_t)is);
	}
	if (verbose >= 3) fprintf(stderr, "[M::mem_pestat] # candidate unique pairs for (FF, FR, RF, RR): (%zu, %zu, %zu, %zu)\n", isize[0].size(), isize[1].size(), isize[2].size(), isize[3].size());
	for (int d = 0; d < 4; ++d) {
		PeStat *r = &pes[d];
		std::vector<uint64_t> &q = isize[d];
		if (q.size() < 10) { r->failed = 1; continue; }
		introsort(q.size(), q.data(), [](uint64_t a, uint64_t b) { return a < b; });
		int p25 = (int)q[(int)(.25 * q.size() + .499)], p75 = (int)q[(int)(.75 * q.size() + .499)], x = 0;
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		r->avg = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) { r->avg += v; ++x; }
		r->avg /= x;
		r->std = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg);
		r->std = sqrt(r->std / x);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
	}
	size_t max = 0;
	for (int d = 0; d < 4; ++d) max = max > isize[d].size() ? max : isize[d].size();
	for (int d = 0; d < 4; ++d) if (pes[d].failed == 0 && isize[d].size() < (int)max * 0.05) pes[d].failed = 1;
}

static int matesw(const Opt &o, const Ref &ref, const PeStat pes[4], const Reg *a, int l_ms, const uint8_t *ms, RegV &ma)   // bwamem_pair.c:137
{
	const int64_t l_pac = ref.l_pac;
	int skip[4], n = 0, rid = -1;
	for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (size_t i = 0; i < ma.a.size(); ++i) {
		int64_t dist;
		int r = infer_dir(l_pac, a->rb, ma.a[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (int r = 0; r < 4; ++r) {
		if (skip[r]) continue;
		const int is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
		std::vector<uint8_t> seq(ms, ms + l_ms), refseq;
		if (is_rev) for (int i = 0; i < l_ms; ++i) seq[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
		int64_t rb, re;
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) fetch_seq(ref, &rb, (rb + re) >> 1, &re, &rid, refseq);
		if (a->rid == rid && re - rb >= o.min_seed_len) {
			int xtra = XSUBO | XSTART | (l_ms * o.a < 250 ? XBYTE : 0) | (o.min_seed_len * o.a);
			Kswr aln = ksw_align2(l_ms, seq.data(), (int)(re - rb), refseq.data(), o.mat, o.o_del, o.e_del, o.o_ins, o.e_ins, xtra);
			Reg b;
			memset(&b, 0, sizeof b);
			if (aln.score >= o.min_seed_len && aln.qb >= 0) {
				b.rid = a->rid; b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score; b.csub = aln.score2; b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				ma.a.push_back(b);
				size_t i;
				for (i = 0; i < ma.a.size() - 1; ++i) if (ma.a[i].score < b.score) break;
				size_t tmp = i;
				for (i = ma.a.size() - 1; i > tmp; --i) ma.a[i] = ma.a[i - 1];
				ma.a[i] = b;
			}
			++n;
		}
		if (n) ma.a.resize(sort_dedup_nopatch(o, (int)ma.a.size(), ma.a.data()));
	}
	return n;
}

struct P64 { uint64_t x, y; };
static int mem_pair(const Opt &o, const Ref &ref, const PeStat pes[4], RegV a[2], int id, int *sub, int *n_sub, int z[2], int n_pri[2])   // bwamem_pair.c:208
{
	std::vector<P64> v, u;
	int y[4], ret;
	const int64_t l_pac = ref.l_pac;
	auto lt = [](const P64 &p, const P64 &q) { return p.x < q.x || (p.x == q.x && p.y < q.y); };
	for (int r = 0; r < 2; ++r)
		for (int i = 0; i < n_pri[r]; ++i) {
			P64 key;
			Reg *e = &a[r].a[i];
			key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;
			key.x = (uint64_t)e->rid << 32 | (key.x - ref.bns->anns[e->rid].offset);
			key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
			v.push_back(key);
		}
	introsort(v.size(), v.data(), lt);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (int i = 0; i < (int)v.size(); ++i) {
		for (int r = 0; r < 2; ++r) {
			int dir = r << 1 | (v[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | ((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (int k = y[which]; k >= 0; --k) {
				if ((int)(v[k].y & 3) != which) continue;
				int64_t dist = (int64_t)v[i].x - v[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				double ns = (dist - pes[dir].avg) / pes[dir].std;
				int q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * o.a + .499);
				if (q < 0) q = 0;
				P64 p;
				p.y = (uint64_t)k << 32 | i;
				p.x = (uint64_t)q << 32 | (hash_64(p.y ^ id << 8) & 0xffffffffU);
				u.push_back(p);
			}
		}
		y[v[i].y & 3] = i;
	}
	if (!u.empty()) {
		int tmp = o.a + o.b;
		tmp = tmp > o.o_del + o.e_del ? tmp : o.o_del + o.e_del;
		tmp = tmp > o.o_ins + o.e_ins ? tmp : o.o_ins + o.e_ins;
		introsort(u.size(), u.data(), lt);
		int i = (int)(u.back().y >> 32), k = (int)(u.back().y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
		ret = (int)(u.back().x >> 32);
		*sub = u.size() > 1 ? (int)(u[u.size() - 2].x >> 32) : 0;
		*n_sub = 0;
		for (long t = (long)u.size() - 2; t >= 0; --t) if (*sub - (int)(u[t].x >> 32) <= tmp) ++*n_sub;
	} else { ret = 0; *sub = 0; *n_sub = 0; }
	return ret;
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

static int sam_pe(const Opt &o, const Ref &ref, const PeStat pes[4], uint64_t id, Read s[2], RegV a[2], char *sam[2])   // bwamem_pair.c:276
{
	int n = 0, z[2], oo, subo, n_sub, extra_flag = 1, n_pri[2];
	Aln h[2];
	bool no_pairing = false;
	if (!(o.flag & BWAHIP_F_NO_RESCUE)) {
		RegV b[2];
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < a[i].a.size(); ++j)
				if (a[i].a[j].score >= a[i].a[0].score - o.pen_unpaired) b[i].a.push_back(a[i].a[j]);
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < b[i].a.size() && (int)j < o.max_matesw; ++j)
				n += matesw(o, ref, pes, &b[i].a[j], s[!i].l_seq, (const uint8_t*)s[!i].seq, a[!i]);
	}
	n_pri[0] = mark_primary_se(o, (int)a[0].a.size(), a[0].a.data(), (int64_t)(id << 1 | 0));
	n_pri[1] = mark_primary_se(o, (int)a[1].a.size(), a[1].a.data(), (int64_t)(id << 1 | 1));
	if (o.flag & BWAHIP_F_PRIMARY5) { reorder_primary5(o.T, a[0]); reorder_primary5(o.T, a[1]); }
	if (o.flag & BWAHIP_F_NOPAIRING) no_pairing = true;
	if (!no_pairing && n_pri[0] && n_pri[1] && (oo = mem_pair(o, ref, pes, a, (int)id, &subo, &n_sub, z, n_pri)) > 0) {
		int is_multi[2], q_pe, score_un, q_se[2];
		for (int i = 0; i < 2; ++i) {
			int j;
			for (j = 1; j < n_pri[i]; ++j) if (a[i].a[j].secondary < 0 && a[i].a[j].score >= o.T) break;
			is_multi[i] = j < n_pri[i] ? 1 : 0;
		}
		if (is_multi[0] || is_multi[1]) no_pairing = true;
		else {
			score_un = a[0].a[0].score + a[1].a[0].score - o.pen_unpaired;
			subo = subo > score_un ? subo : score_un;
			q_pe = RAW_MAPQ(oo - subo, o.a);
			if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
			if (q_pe < 0) q_pe = 0;
			if (q_pe > 60) q_pe = 60;
			q_pe = (int)(q_pe * (1. - .5 * (a[0].a[0].frac_rep + a[1].a[0].frac_rep)) + .499);
			if (oo > score_un) {
				Reg *c[2] = { &a[0].a[z[0]], &a[1].a[z[1]] };
				for (int i = 0; i < 2; ++i) {
					if (c[i]->secondary >= 0) { c[i]->sub = a[i].a[c[i]->secondary].score; c[i]->secondary = -2; }
					q_se[i] = approx_mapq_se(o, c[i]);
				}
				q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
				q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
				extra_flag |= 2;
				q_se[0] = q_se[0] < RAW_MAPQ(c[0]->score - c[0]->csub, o.a) ? q_se[0] : RAW_MAPQ(c[0]->score - c[0]->csub, o.a);
				q_se[1] = q_se[1] < RAW_MAPQ(c[1]->score - c[1]->csub, o.a) ? q_se[1] : RAW_MAPQ(c[1]->score - c[1]->csub, o.a);
			} else {
				z[0] = z[1] = 0;
				q_se[0] = approx_mapq_se(o, &a[0].a[0]);
				q_se[1] = approx_mapq_se(o, &a[1].a[0]);
			}
			for (int i = 0; i < 2; ++i) {
				int k = a[i].a[z[i]].secondary_all;
				if (k >= 0 && k < n_pri[i]) {
					for (size_t j = 0; j < a[i].a.size(); ++j)
						if (a[i].a[j].secondary_all == k || (int)j == k) a[i].a[j].secondary_all = z[i];
					a[i].a[z[i]].secondary_all = -1;
				}
			}
			std::vector<std::string> XA[2]; std::vector<char> has[2]; bool any[2] = { false, false };
			if (!(o.flag & BWAHIP_F_ALL)) for (int i = 0; i < 2; ++i) gen_alt(o, ref, a[i], s[i].l_seq, s[i].seq, XA[i], has[i], any[i]);
			std::vector<Aln> aa[2];
			Aln g[2];
			for (int i = 0; i < 2; ++i) {
				h[i] = reg2aln(o, ref, s[i].l_seq, s[i].seq, &a[i].a[z[i]]);
				h[i].mapq = (uint32_t)q_se[i] & 0xff;
				h[i].flag |= 0x40 << i | extra_flag;
				if (any[i] && has[i][z[i]]) { h[i].XA = XA[i][z[i]]; h[i].has_XA = true; }
				aa[i].push_back(h[i]);
				if (n_pri[i] < (int)a[i].a.size()) {
					Reg *p = &a[i].a[n_pri[i]];
					if (p->score < o.T || p->secondary >= 0 || !p->is_alt) continue;
					g[i] = reg2aln(o, ref, s[i].l_seq, s[i].seq, p);
					g[i].flag |= 0x800 | 0x40 << i | extra_flag;
					if (any[i] && has[i][n_pri[i]]) { g[i].XA = XA[i][n_pri[i]]; g[i].has_XA = true; }
					aa[i].push_back(g[i]);
				}
			}
			std::string str;
			for (size_t i = 0; i < aa[0].size(); ++i) aln2sam(o, ref, str, s[0], (int)aa[0].size(), aa[0].data(), (int)i, &h[1]);
			sam[0] = dup_sam(str); str.clear();
			for (size_t i = 0; i < aa[1].size(); ++i) aln2sam(o, ref, str, s[1], (int)aa[1].size(), aa[1].data(), (int)i, &h[0]);
			sam[1] = dup_sam(str);
			return n;
		}
	} else no_pairing = true;
	// no_pairing: (bwamem_pair.c:397-418)
	for (int i = 0; i < 2; ++i) {
		int which = -1;
		if (!a[i].a.empty()) {
			if (a[i].a[0].score >= o.T) which = 0;
			else if (n_pri[i] < (int)a[i].a.size() && a[i].a[n_pri[i]].score >= o.T) which = n_pri[i];
		}
		h[i] = which >= 0 ? reg2aln(o, ref, s[i].l_seq, s[i].seq, &a[i].a[which]) : reg2aln(o, ref, s[i].l_seq, s[i].seq, 0);
	}
	if (!(o.flag & BWAHIP_F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {
		int64_t dist;
		int d = infer_dir(ref.l_pac, a[0].a[0].rb, a[1].a[0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	reg2sam(o, ref, s[0], a[0], 0x41 | extra_flag, &h[1], &sam[0]);
	reg2sam(o, ref, s[1], a[1], 0x81 | extra_flag, &h[0], &sam[1]);
	return n;
}

} // namespace hf

// ================================================================ C ABI: bwahip_process_seqs == mem_process_seqs (bwamem.c:1215)
// Host finalisation after the GPU hot path (the gpu_final = 0 / gpu_pair = 0 knobs of the tests).
extern "C" int bwahip_process_seqs_host(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0)
{
	if (!ctx || !opt || n < 0 || (n && !seqs)) return BWAHIP_EINVAL;
	const bool pe = (opt->flag & BWAHIP_F_PE) != 0;
	if (pe && (n & 1)) return BWAHIP_EINVAL;
	const bwahip_bns_t *bns = bwahip_bns(ctx);
	const uint8_t *pac = bwahip_pac(ctx);
	if (!bns || !pac) { fprintf(stderr, "[bwahip] process_seqs needs a host copy of .pac/.ann (context was built from device arrays)\n"); return BWAHIP_EINVAL; }
	// phase 1: the hot path on the GPU (kt_for(worker1), bwamem.c:1232)
	std::vector<bwahip_alnreg_v> regs_c(n);
	int rc = bwahip_align_batch(ctx, opt, n, seqs, regs_c.data());
	if (rc) return rc;
	std::vector<hf::RegV> regs(n);
	par_for_chunks(n, opt->n_threads, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) { regs[i].a.assign(regs_c[i].a, regs_c[i].a + regs_c[i].n); free(regs_c[i].a); }
	});
	hf::Ref ref = { bns, pac, bns->l_pac, bwahip_ctx_rg_id(ctx) };
	// phase 2 (serial): insert-size statistics (bwamem.c:1236-1239)
	bwahip_pestat_t pes[4];
	memset(pes, 0, sizeof pes);
	if (pe) {
		if (pes0) memcpy(pes, pes0, sizeof pes);
		else hf::pestat(*opt, bns->l_pac, n, regs.data(), pes, 0);
	}
	// phase 3: worker2 on host threads (bwamem.c:1197-1213, 1240)
	const int n_items = pe ? n >> 1 : n;
	std::atomic<int> next(0), bad(0);
	auto work = [&]() {
		for (;;) {
			const int i0 = next.fetch_add(64);                   // 64 items per grab: one shared counter, hundreds of threads
			if (i0 >= n_items) break;
			for (int i = i0; i < n_items && i < i0 + 64; ++i)
			if (!pe) {
				hf::Read s = { seqs[i].name, seqs[i].comment, seqs[i].qual, seqs[i].seq, seqs[i].l_seq };
				hf::mark_primary_se(*opt, (int)regs[i].a.size(), regs[i].a.data(), n_processed + i);
				if (opt->flag & BWAHIP_F_PRIMARY5) hf::reorder_primary5(opt->T, regs[i]);
				hf::reg2sam(*opt, ref, s, regs[i], 0, 0, &seqs[i].sam);
			} else {
				hf::Read s[2] = { { seqs[i<<1].name, seqs[i<<1].comment, seqs[i<<1].qual, seqs[i<<1].seq, seqs[i<<1].l_seq },
				                  { seqs[i<<1|1].name, seqs[i<<1|1].comment, seqs[i<<1|1].qual, seqs[i<<1|1].seq, seqs[i<<1|1].l_seq } };
				if (strcmp(s[0].name, s[1].name) != 0) { bad = 1; continue; }   // err_fatal in the reference (bwamem_pair.c:386)
				char *sam[2] = { 0, 0 };
				hf::sam_pe(*opt, ref, pes, (uint64_t)((n_processed >> 1) + i), s, &regs[i << 1], sam);
				seqs[i<<1].sam = sam[0]; seqs[i<<1|1].sam = sam[1];
			}
		}
	};
	const int nt = opt->n_threads > 1 ? opt->n_threads : 1;
	if (nt == 1) work();
	else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto &t : th) t.join(); }
	if (bad) { fprintf(stderr, "[bwahip] paired reads have different names\n"); return BWAHIP_EINVAL; }
	return 0;
}

