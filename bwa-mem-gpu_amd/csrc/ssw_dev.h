// Striped local alignment of the reference (ksw_u8 ksw.c:111-230, ksw_i16 ksw.c:232-341, ksw_align2 ksw.c:343-365),
// emulated lane for lane by a GROUP of P lanes of a wavefront: P = 16 stands for the sixteen unsigned bytes of an
// __m128i (KSW_XBYTE), P = 8 for its eight signed words.  The emulation is literal on purpose: the SSE2 code's results
// are visible in the SAM (scores, end points, second-best score), and they depend on the striping itself -- E is fed from
// the H that lacks the cross-lane lazy-F correction (ksw.c:176), saturation differs between the byte and word kernels,
// and the lazy-F loop stops on a whole-vector test.  One lane owns one SSE lane: its cells H/E of every segment live in
// LDS at [segment * P + lane] (no other lane ever touches them), the byte/word shifts of the vector code
// (_mm_slli_si128) are DPP row shifts, the horizontal maxima are DPP butterflies, the movemask tests are ballots.
// A wavefront therefore runs 4 (bytes) or 8 (words) independent alignments side by side; groups diverge freely.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ssw {

struct Res { int score, te, qe, score2, te2, tb, qb; };       // kswr_t, ksw.h:42-48

enum { XBYTE = 0x10000, XSTOP = 0x20000, XSUBO = 0x40000, XSTART = 0x80000 };   // ksw.h:30-33

// LDS (or global) working set of one group.  prof: 5 rows of slen*P scores (ksw_qinit, ksw.c:60-110): bytes for P = 16
// (score + shift), signed bytes for P = 8.  H0/H1/E/Hmax: slen*P 16-bit cells each.  colmax: one entry per target
// column (only read when a second-best score is asked for).
struct Work { int8_t *prof; int16_t *H0, *H1, *E, *Hmax; uint16_t *colmax; uint8_t *colmax8 = nullptr; };   // colmax8: the column maxima as bytes (byte kernel only: they are <= 255) instead of colmax
__host__ __device__ constexpr size_t work_bytes(int P, int qlen_max) { return (size_t)((qlen_max + P - 1) / P) * P * (5 + 4 * 2); }

template <int P> __device__ __forceinline__ int gl_of(int lane) { return lane & (P - 1); }

// _mm_slli_si128(v, one element): lane l of the group receives lane l-1, lane 0 receives 0
template <int P> __device__ __forceinline__ int shift_up(int v, int gl)
{
	const int s = __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
	return gl == 0 ? 0 : s;
}
// horizontal maximum over the group (all lanes receive it): xor butterfly on DPP
template <int P> __device__ __forceinline__ int group_max(int v)
{
	int o;
	if (P == 16) { o = __builtin_amdgcn_update_dpp(v, v, 0x140 /* row_mirror */, 0xf, 0xf, false); v = v > o ? v : o; }
	o = __builtin_amdgcn_update_dpp(v, v, 0x141 /* row_half_mirror */, 0xf, 0xf, false); v = v > o ? v : o;
	o = __builtin_amdgcn_update_dpp(v, v, 0x4e /* quad_perm 2,3,0,1 */, 0xf, 0xf, false); v = v > o ? v : o;
	o = __builtin_amdgcn_update_dpp(v, v, 0xb1 /* quad_perm 1,0,3,2 */, 0xf, 0xf, false); v = v > o ? v : o;
	return v;
}
template <int P> __device__ __forceinline__ int group_min(int v) { return -group_max<P>(-v); }
// movemask != 0 over the group; every lane of the group must be active
template <int P> __device__ __forceinline__ bool group_any(bool p, int lane)
{
	const unsigned long long m = __ballot(p);
	return ((m >> (lane & ~(P - 1))) & ((1ull << P) - 1)) != 0;
}

// _mm_subs_epu8 / _mm_subs_epu16 on values held in 32-bit registers: both operands are non-negative wherever this is called (cell values after
// the maxima with E / F, gap penalties), so the unsigned saturating subtraction -- one v_sub_u32 with the clamp bit -- is max(a - b, 0)
__device__ __forceinline__ int sat_sub_u(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }

// Sequence access: element i of a byte string read with a stride, optionally with its first `rev_n` elements reversed
// (revseq of ksw.c:358 on a prefix, the rest untouched).
struct SeqView {
	const uint8_t *p; int stride, rev_n; int nib = 0;            // nib: two elements per byte (element 2k in the low half), stride 1
	__device__ __forceinline__ int at(int i) const
	{
		const int k = i < rev_n ? rev_n - 1 - i : i;
		if (nib) return p[k >> 1] >> ((k & 1) << 2) & 15;
		return p[k * stride];
	}
};

// ksw_qinit (ksw.c:60-110): shift / max of the matrix; profile in the striped order.  Group-collective.
template <int P>
__device__ __forceinline__ void qinit(const Work &w, int gl, int qlen, const SeqView &q, const int8_t *mat, int &slen_, int &shift_, int &max_)
{
	const int slen = (qlen + P - 1) / P;
	int mn = 127, mx = 0;
	for (int a = 0; a < 25; ++a) { const int v = mat[a]; if (v < (int)(int8_t)mn) mn = v & 0xff; if (v > (int)(int8_t)mx) mx = v & 0xff; }
	// (the reference keeps the running minimum / maximum in uint8_t fields: q->shift, q->mdiff)
	max_ = mx;
	shift_ = (256 - mn) & 0xff;
	for (int a = 0; a < 5; ++a)
		for (int j = 0; j < slen; ++j) {
			const int k = j + gl * slen;                              // query position of segment j in lane gl
			const int v = k >= qlen ? 0 : mat[a * 5 + q.at(k)];
			w.prof[(a * slen + j) * P + gl] = (int8_t)(P == 16 ? ((v + shift_) & 0xff) : v);
		}
	slen_ = slen;
}

// One pass of ksw_u8 (P = 16) / ksw_i16 (P = 8).  minsc / endsc as decoded from xtra (0x10000 = off).  WANT_QE: also
// keep Hmax and return qe.  Returns score/te/qe and score2/te2 (the latter two only when minsc is on).  Group-collective.
template <int P, bool WANT_QE>
__device__ __forceinline__ Res pass(const Work &w, int lane, int slen, int shift, int qmax, int tlen, const SeqView &t,
                                    int o_del, int e_del, int o_ins, int e_ins, int minsc, int endsc)
{
	constexpr bool is8 = P == 16;
	const int gl = gl_of<P>(lane);
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	if (is8) { oe_del &= 0xff; oe_ins &= 0xff; e_del &= 0xff; e_ins &= 0xff; }      // _mm_set1_epi8
	else { oe_del &= 0xffff; oe_ins &= 0xffff; e_del &= 0xffff; e_ins &= 0xffff; }  // _mm_set1_epi16
	int16_t *H0 = w.H0, *H1 = w.H1, *E = w.E, *Hmax = w.Hmax;
	for (int j = 0; j < slen; ++j) { E[j * P + gl] = 0; H0[j * P + gl] = 0; if (WANT_QE) Hmax[j * P + gl] = 0; }
	int gmax = 0, te = -1, n_cols = 0;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *S = w.prof + (size_t)t.at(i) * slen * P;
		int f = 0, mxv = 0;
		int h = slen > 0 ? shift_up<P>((int)H0[(slen - 1) * P + gl], gl) : 0;
		for (int j = 0; j < slen; ++j) {
			int e = E[j * P + gl];
			if (is8) { h += (int)(uint8_t)S[j * P + gl]; h = h > 255 ? 255 : h; h = sat_sub_u(h, shift); }
			else { h += (int)S[j * P + gl]; h = h > 32767 ? 32767 : h < -32768 ? -32768 : h; }
			h = h > e ? h : e;
			h = h > f ? h : f;
			mxv = mxv > h ? mxv : h;
			H1[j * P + gl] = (int16_t)h;
			e = sat_sub_u(e, e_del);
			int tt = sat_sub_u(h, oe_del);
			e = e > tt ? e : tt;
			E[j * P + gl] = (int16_t)e;
			f = sat_sub_u(f, e_ins);
			tt = sat_sub_u(h, oe_ins);
			f = f > tt ? f : tt;
			h = H0[j * P + gl];
		}
		// lazy-F loop (ksw.c:179-190 / 287-297): 16 rounds at most in both kernels
		for (int k = 0; k < 16; ++k) {
			f = shift_up<P>(f, gl);
			bool stop = false;
			for (int j = 0; j < slen; ++j) {
				int hh = H1[j * P + gl];
				hh = hh > f ? hh : f;
				H1[j * P + gl] = (int16_t)hh;
				hh = sat_sub_u(hh, oe_ins);
				f = sat_sub_u(f, e_ins);
				if (!group_any<P>(f > hh, lane)) { stop = true; break; }
			}
			if (stop) break;
		}
		const int imax = group_max<P>(mxv);
		if (gl == 0) { if (w.colmax8) w.colmax8[i] = (uint8_t)imax; else if (w.colmax) w.colmax[i] = (uint16_t)imax; }
		n_cols = i + 1;
		if (imax > gmax) {
			gmax = imax; te = i;
			if (WANT_QE) for (int j = 0; j < slen; ++j) Hmax[j * P + gl] = H1[j * P + gl];
			if (is8 ? (gmax + shift >= 255 || gmax >= endsc) : gmax >= endsc) break;
		}
		int16_t *tmp = H1; H1 = H0; H0 = tmp;
	}
	Res r = { 0, -1, -1, -1, -1, -1, -1 };                        // g_defr, ksw.c:49
	r.score = is8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!is8 || r.score != 255) {
		if (WANT_QE) {
			// smallest query index among the cells of the best column that hold its maximum (ksw.c:211-216 / 318-322)
			int best = -1;
			for (int j = 0; j < slen; ++j) { const int v = (int)(uint16_t)Hmax[j * P + gl]; best = best > v ? best : v; }
			best = group_max<P>(best);
			int qe = 1 << 30;
			for (int j = 0; j < slen; ++j) if ((int)(uint16_t)Hmax[j * P + gl] == best) { const int k = j + gl * slen; qe = qe < k ? qe : k; }
			qe = group_min<P>(qe);
			r.qe = slen > 0 ? qe : -1;
		}
		if (minsc < 0x10000 && (w.colmax || w.colmax8)) {
			// the b[] list of ksw.c:194-204 replayed from the column maxima, then the scan of ksw.c:217-225; every lane
			// of the group runs it (same values everywhere)
			__threadfence_block();                                    // colmax was written by lane 0 of the group
			const int span = (r.score + qmax - 1) / qmax, low = te - span, high = te + span;
			int cur_m = -1, cur_i = -1;
			bool have = false;
			for (int i = 0; i < n_cols; ++i) {
				const int m = w.colmax8 ? (int)w.colmax8[i] : (int)w.colmax[i];
				if (m < minsc) continue;
				if (!have || cur_i + 1 != i) {
					if (have && (cur_i < low || cur_i > high) && cur_m > r.score2) { r.score2 = cur_m; r.te2 = cur_i; }
					cur_m = m; cur_i = i; have = true;
				} else if (cur_m < m) { cur_m = m; cur_i = i; }
			}
			if (have && (cur_i < low || cur_i > high) && cur_m > r.score2) { r.score2 = cur_m; r.te2 = cur_i; }
		}
	}
	return r;
}

// The same pass with the H / E / Hmax cells of a lane in REGISTERS (segments fully unrolled up to SLEN): the DP column is a
// chain of dependent steps, and with the cells in LDS every step waits for an LDS round trip (~1 900 cycles per column
// measured with 10 segments); in registers only the profile row comes from LDS, and its loads do not depend on the chain.
// Exactly the arithmetic of pass<>; w.H0/H1/E are not touched, w.Hmax receives the best column at the end (for callers
// that want it; qe is computed here).
// FULL: every lane's query has exactly SLEN segments (slen == SLEN in the whole wavefront -- reads of one length): the per-segment tests
// `j < slen` are compile-time true and the unrolled loops have no branches around their bodies
template <int P, int SLEN, bool FULL = false>
__device__ __forceinline__ Res pass_reg(const Work &w, int lane, int slen, int shift, int qmax, int tlen, const SeqView &t,
                                        int o_del, int e_del, int o_ins, int e_ins, int minsc, int endsc)
{
	constexpr bool is8 = P == 16;
	const int gl = gl_of<P>(lane);
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	if (is8) { oe_del &= 0xff; oe_ins &= 0xff; e_del &= 0xff; e_ins &= 0xff; }
	else { oe_del &= 0xffff; oe_ins &= 0xffff; e_del &= 0xffff; e_ins &= 0xffff; }
	int H[SLEN], E[SLEN], Hm[SLEN];
#pragma unroll
	for (int j = 0; j < SLEN; ++j) { H[j] = 0; E[j] = 0; Hm[j] = 0; }
	int gmax = 0, te = -1, n_cols = 0;
	// the profile row of column i + 1 is fetched while column i is computed: the byte loads may alias the column-maximum store at the end of
	// a column as far as the compiler knows, so left to it every column began with two dependent LDS round trips (base, then its row)
	int sv[SLEN];
	{
		const int8_t *S = w.prof + (size_t)(tlen > 0 ? t.at(0) : 0) * slen * P;
#pragma unroll
		for (int j = 0; j < SLEN; ++j) sv[j] = (FULL || j < slen) ? (is8 ? (int)(uint8_t)S[j * P + gl] : (int)S[j * P + gl]) : 0;
	}
	for (int i = 0; i < tlen; ++i) {
		int svn[SLEN];
		{
			const int8_t *S = w.prof + (size_t)t.at(i + 1 < tlen ? i + 1 : i) * slen * P;
#pragma unroll
			for (int j = 0; j < SLEN; ++j) svn[j] = (FULL || j < slen) ? (is8 ? (int)(uint8_t)S[j * P + gl] : (int)S[j * P + gl]) : 0;
		}
		int f = 0, mxv = 0, last = 0;
#pragma unroll
		for (int j = 0; j < SLEN; ++j) if (FULL ? j == SLEN - 1 : j == slen - 1) last = H[j];
		int h = shift_up<P>(last, gl);
#pragma unroll
		for (int j = 0; j < SLEN; ++j) {
			if ((FULL || j < slen)) {
				int e = E[j];
				if (is8) { h += sv[j]; h = h > 255 ? 255 : h; h = sat_sub_u(h, shift); }
				else { h += sv[j]; h = h > 32767 ? 32767 : h < -32768 ? -32768 : h; }
				h = h > e ? h : e;
				h = h > f ? h : f;
				mxv = mxv > h ? mxv : h;
				const int h_old = H[j];
				H[j] = h;
				e = sat_sub_u(e, e_del);
				int tt = sat_sub_u(h, oe_del);
				E[j] = e > tt ? e : tt;
				f = sat_sub_u(f, e_ins);
				tt = sat_sub_u(h, oe_ins);
				f = f > tt ? f : tt;
				h = h_old;
			}
		}
		bool stop = false;
		for (int k = 0; k < 16 && !stop; ++k) {                     // lazy-F (ksw.c:179-190 / 287-297)
			f = shift_up<P>(f, gl);
#pragma unroll
			for (int j = 0; j < SLEN; ++j) {
				if ((FULL || j < slen) && !stop) {
					int hh = H[j];
					hh = hh > f ? hh : f;
					H[j] = hh;
					hh = sat_sub_u(hh, oe_ins);
					f = sat_sub_u(f, e_ins);
					if (!group_any<P>(f > hh, lane)) stop = true;
				}
			}
		}
		const int imax = group_max<P>(mxv);
		if (gl == 0) { if (w.colmax8) w.colmax8[i] = (uint8_t)imax; else if (w.colmax) w.colmax[i] = (uint16_t)imax; }
		n_cols = i + 1;
		if (imax > gmax) {
			gmax = imax; te = i;
#pragma unroll
			for (int j = 0; j < SLEN; ++j) Hm[j] = H[j];
			if (is8 ? (gmax + shift >= 255 || gmax >= endsc) : gmax >= endsc) break;
		}
#pragma unroll
		for (int j = 0; j < SLEN; ++j) sv[j] = svn[j];
	}
	Res r = { 0, -1, -1, -1, -1, -1, -1 };
	r.score = is8 ? (gmax + shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!is8 || r.score != 255) {
		int best = -1;
#pragma unroll
		for (int j = 0; j < SLEN; ++j) if ((FULL || j < slen)) { const int v = Hm[j] & 0xffff; best = best > v ? best : v; }
		best = group_max<P>(best);
		int qe = 1 << 30;
#pragma unroll
		for (int j = 0; j < SLEN; ++j) if ((FULL || j < slen) && (Hm[j] & 0xffff) == best) { const int k = j + gl * slen; qe = qe < k ? qe : k; }
		qe = group_min<P>(qe);
		r.qe = slen > 0 ? qe : -1;
		if (minsc < 0x10000 && (w.colmax || w.colmax8)) {
			__threadfence_block();
			const int span = (r.score + qmax - 1) / qmax, low = te - span, high = te + span;
			int cur_m = -1, cur_i = -1;
			bool have = false;
			for (int i = 0; i < n_cols; ++i) {
				const int m = w.colmax8 ? (int)w.colmax8[i] : (int)w.colmax[i];
				if (m < minsc) continue;
				if (!have || cur_i + 1 != i) {
					if (have && (cur_i < low || cur_i > high) && cur_m > r.score2) { r.score2 = cur_m; r.te2 = cur_i; }
					cur_m = m; cur_i = i; have = true;
				} else if (cur_m < m) { cur_m = m; cur_i = i; }
			}
			if (have && (cur_i < low || cur_i > high) && cur_m > r.score2) { r.score2 = cur_m; r.te2 = cur_i; }
		}
	}
	return r;
}

// ksw_align2 (ksw.c:343-365).  q / t are plain (unreversed) views; `mat` 5x5.  Group-collective; the caller provides
// the group's working set and separates consecutive calls that reuse it with a wavefront barrier.
// SLEN > 0: cells in registers (pass_reg) when the query has at most SLEN segments, else the LDS version.
template <int P, int SLEN = 0>
__device__ __forceinline__ Res align2(const Work &w, int lane, int qlen, const uint8_t *q, int qstride, int tlen, const uint8_t *t, int tstride,
                                      const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, bool t_nib = false)
{
	const int gl = gl_of<P>(lane);
	int slen, shift, qmax;
	SeqView qv = { q, qstride, 0 }, tv = { t, tstride, 0 };
	tv.nib = t_nib ? 1 : 0;
	qinit<P>(w, gl, qlen, qv, mat, slen, shift, qmax);
	const int minsc = (xtra & XSUBO) ? xtra & 0xffff : 0x10000, endsc = (xtra & XSTOP) ? xtra & 0xffff : 0x10000;
	const bool in_regs = SLEN > 0 && slen <= SLEN;
	const bool full = SLEN > 0 && __ballot(slen != SLEN) == 0;      // (wave-uniform; all lanes of the wavefront are here)
	Res r = full ? pass_reg<P, SLEN ? SLEN : 1, true>(w, lane, slen, shift, qmax, tlen, tv, o_del, e_del, o_ins, e_ins, minsc, endsc)
	      : in_regs ? pass_reg<P, SLEN ? SLEN : 1>(w, lane, slen, shift, qmax, tlen, tv, o_del, e_del, o_ins, e_ins, minsc, endsc)
	                : pass<P, true>(w, lane, slen, shift, qmax, tlen, tv, o_del, e_del, o_ins, e_ins, minsc, endsc);
	if ((xtra & XSTART) == 0 || ((xtra & XSUBO) && r.score < (xtra & 0xffff))) return r;
	if (P == 16 && r.score == 255) return r;                      // qe unknown: the reference reads out of bounds here; unreachable (score <= qlen*a < 250)
	// second pass on the reversed prefixes to find the start (ksw.c:356-363); it scans tlen columns, not te+1
	qv.rev_n = r.qe + 1; tv.rev_n = r.te + 1;
	Work w2 = w; w2.colmax = nullptr; w2.colmax8 = nullptr;
	qinit<P>(w2, gl, r.qe + 1, qv, mat, slen, shift, qmax);
	const Res rr = in_regs ? pass_reg<P, SLEN ? SLEN : 1>(w2, lane, slen, shift, qmax, tlen, tv, o_del, e_del, o_ins, e_ins, 0x10000, r.score & 0xffff)
	                       : pass<P, true>(w2, lane, slen, shift, qmax, tlen, tv, o_del, e_del, o_ins, e_ins, 0x10000, r.score & 0xffff);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

} // namespace ssw
