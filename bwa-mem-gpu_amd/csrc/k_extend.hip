// K4/K5 -- seed extension and region clean-up: mem_chain2aln (bwamem.c:639-793) with ksw_extend2
// (ksw.c:380-479), then mem_sort_dedup_patch (bwamem.c:444-496) with mem_patch_reg (bwamem.c:413) and
// the score-only banded global alignment it needs (bwa_gen_cigar2 bwa.c:261-307, ksw_global2
// ksw.c:504-584).  One read per wavefront (workgroup = 1 wave, 64 lanes).
//
// DP on a wavefront.  Both DPs feed E and F from M (= H(i-1,j-1)+s), not from H (ksw.c:439-447,
// 556-564), so a row has no serial dependency except F, and F is a max-plus prefix scan:
//     F(i,j) = max_{k<j} ( max(M_k - oe_ins, 0) + k*e_ins ) - (j-1)*e_ins .
// Each lane owns 1..CPL adjacent query columns in registers -- the fewest that hold the flank -- (H and E rows
// never touch memory); one row = local work + one wavefront exclusive max-scan + a max reduction (row max with
// last column) + a ballot (first / last non-zero cell for the band trimming of ksw.c:466-469); the band and all
// row control are wavefront-uniform and run on the scalar unit.  Rows stay sequential, which keeps band
// trimming, z-drop (ksw.c:458-464) and every tie rule exactly as in the reference.
// Integer DP, no dense contraction: MFMA is not applicable.
//
// Two hand-offs keep one read's serial work from setting the duration of the launch: k_extend_spec extends the best
// seed of every chain of many-chain reads ahead of time (one wavefront per chain), and dedup sorts long region
// lists with a wavefront rank sort whenever no two keys tie (otherwise with the exact one-lane introsort).
//
// The per-seed control flow (containment skip bwamem.c:678-713, band doubling bwamem.c:730-741,
// clip-vs-to-end choice bwamem.c:743-749) is scalar and uniform across the wavefront; tests over
// lists (previous regions, overlapping seeds) use one lane per element and a ballot.
#include "bwahip_internal.h"
#include "wave_dev.h"
#include "regsort_dev.h"

#ifndef KEXT_W3
#define KEXT_W3 7                                           // wavefronts per SIMD the extension kernels for reads below 192 bases are compiled for
#endif
namespace {
using namespace wv;

constexpr int MAXQ = BWAHIP_MAX_READ_LEN;                    // query / columns
constexpr int MAXT = BWAHIP_MAX_READ_LEN + 2 * 200 + 1024;   // reference window kept in LDS: l_query + 2*max_gap(<= 2w at the default w) + diagonal drift; larger windows go to k_extend_big

// per-wavefront work counters (all uniform; lane 0 adds them to the launch counters at the end)
struct Work { unsigned long long cells; unsigned rows1, rowsN; };

// ---------------------------------------------------------------------------------------------------
// ksw_extend2 (ksw.c:380).  q/t live in LDS and are read with a stride (+1 / -1) so the left extension
// can run on the reversed sequences (bwamem.c:725-729) without copying.  Collective over the wavefront.
// ---------------------------------------------------------------------------------------------------
template <int CPL>
__device__ int wave_extend(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen,
                           int w, int end_bonus, int zdrop, int h0, int &qle, int &tle, int &gtle, int &gscore_, int &max_off_,
                           Work &wk)
{
	const int l = lane(), j0 = l * CPL;
	// every argument is wavefront-uniform, but it reaches here through per-lane loads: tell the compiler, so that the
	// band bookkeeping and all control flow of the row loop run on the scalar unit
	qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); w = __builtin_amdgcn_readfirstlane(w);
	end_bonus = __builtin_amdgcn_readfirstlane(end_bonus); zdrop = __builtin_amdgcn_readfirstlane(zdrop); h0 = __builtin_amdgcn_readfirstlane(h0);
	const int oe_del = __builtin_amdgcn_readfirstlane(sw.o_del + sw.e_del), oe_ins = __builtin_amdgcn_readfirstlane(sw.o_ins + sw.e_ins);
	const int e_del = __builtin_amdgcn_readfirstlane(sw.e_del), e_ins = __builtin_amdgcn_readfirstlane(sw.e_ins);
	int qv[CPL], Hs[CPL], E[CPL];
	// first row (ksw.c:396-397) and the query codes of this lane's columns
	const int h1st = h0 > oe_ins ? h0 - oe_ins : 0;
#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int j = j0 + c;
		qv[c] = j < qlen ? q[j * qs] : 4;
		int v = 0;
		if (j == 0) v = h0;
		else if (j == 1) v = h1st;
		else if (j <= qlen && h1st - (j - 2) * e_ins > e_ins) v = h1st - (j - 1) * e_ins;
		Hs[c] = v; E[c] = 0;
	}
	// clamp the band (ksw.c:399-407)
	const int mx = sw.mx;
	int max_ins = div_plus1_trunc(qlen * mx + end_bonus - sw.o_ins, e_ins);
	max_ins = max_ins > 1 ? max_ins : 1;
	w = w < max_ins ? w : max_ins;
	int max_del = div_plus1_trunc(qlen * mx + end_bonus - sw.o_del, e_del);
	max_del = max_del > 1 ? max_del : 1;
	w = w < max_del ? w : max_del;
	int best = h0, best_i = -1, best_j = -1, best_ie = -1, gscore = -1, max_off = 0;
	int beg = 0, end = qlen;
	int scn[CPL];                                                // substitution scores of the next row (prefetched from LDS)
	{
		const int tb0 = tlen > 0 ? t[0] : 4;
#pragma unroll
		for (int c = 0; c < CPL; ++c) scn[c] = sw.mat[tb0 * 5 + qv[c]];
	}
	for (int i = 0; i < tlen; ++i) {
		int scv[CPL];
#pragma unroll
		for (int c = 0; c < CPL; ++c) scv[c] = scn[c];
		{
			const int tbn = i + 1 < tlen ? t[(i + 1) * ts] : 4;   // issued now, consumed next iteration
#pragma unroll
			for (int c = 0; c < CPL; ++c) scn[c] = sw.mat[tbn * 5 + qv[c]];
		}
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		int h1 = 0;
		if (beg == 0) { h1 = h0 - (sw.o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		wk.cells += end > beg ? (unsigned)(end - beg) : 0u;          // wavefront-uniform; lane 0 reports it
		if (CPL == 1) ++wk.rows1; else ++wk.rowsN;
		int M[CPL], u[CPL], P = NEG;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			M[c] = Hs[c] ? Hs[c] + scv[c] : 0;                     // ksw.c:433
			int tI = M[c] - oe_ins; tI = tI > 0 ? tI : 0;
			u[c] = inb ? tI + j * e_ins : NEG;
			P = P > u[c] ? P : u[c];
		}
		int run = wscan_excl_max(P, NEG);
		int h[CPL], key = -1, firstnz = 1 << 30, lastnz = -1;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			const int f = j == beg ? 0 : run - (j - 1) * e_ins;   // F(i,j); for in-band j > beg `run` holds a real value
			int hv = M[c] > E[c] ? M[c] : E[c];
			hv = hv > f ? hv : f;
			h[c] = inb ? hv : 0;
			if (inb) {
				int tD = M[c] - oe_del; tD = tD > 0 ? tD : 0;
				int en = E[c] - e_del; en = en > tD ? en : tD;
				E[c] = en;
				const int k = hv * 1024 + j;                      // row max, last column wins ties (ksw.c:437-438)
				key = key > k ? key : k;
			}
			run = run > u[c] ? run : u[c];
		}
		// shift: eh[j+1].h = H(i,j) (ksw.c:432 p->h = h1), eh[beg].h = first-column value, eh[end] = {h1, 0}
		const int up = __builtin_amdgcn_update_dpp(0, h[CPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
#pragma unroll
		for (int c = CPL - 1; c >= 0; --c) {
			const int j = j0 + c;
			const int left = c == 0 ? up : h[c - 1];
			if (j == beg) Hs[c] = h1;
			else if (j > beg && j <= end) Hs[c] = left;
			if (j == end) E[c] = 0;
			if (j >= beg && j <= end && (Hs[c] != 0 || E[c] != 0)) { firstnz = firstnz < j ? firstnz : j; lastnz = lastnz > j ? lastnz : j; }
		}
		key = wmax(key);
		const int m = key < 0 ? 0 : key >> 10, mj = key < 0 ? -1 : key & 1023;
		if (end == qlen) {                                        // ksw.c:450-453 (j == qlen after the loop)
			int hend = 0;                                           // eh[end].h = H(i, qlen-1): column `end` sits in lane end / CPL
#pragma unroll
			for (int c = 0; c < CPL; ++c) if (end % CPL == c) hend = __builtin_amdgcn_readlane(Hs[c], end / CPL);
			if (end == beg) hend = h1;
			best_ie = gscore > hend ? best_ie : i;
			gscore = gscore > hend ? gscore : hend;
		}
		if (m == 0) break;
		if (m > best) {
			best = m; best_i = i; best_j = mj;
			const int off = mj > i ? mj - i : i - mj;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - best_i > mj - best_j) {
				if (best - m - ((i - best_i) - (mj - best_j)) * e_del > zdrop) break;
			} else {
				if (best - m - ((mj - best_j) - (i - best_i)) * e_ins > zdrop) break;
			}
		}
		// band trimming (ksw.c:466-469); `end` itself is a candidate for the backward scan.  Columns grow with the
		// lane index, so the first / last lane holding a non-zero cell holds the first / last such column.
		const unsigned long long nzm = __ballot(lastnz >= 0);
		int fz = end, lz = -1;
		if (nzm) {
			fz = __builtin_amdgcn_readlane(firstnz, __ffsll((long long)nzm) - 1);
			lz = __builtin_amdgcn_readlane(lastnz, 63 - __clzll((long long)nzm));
			fz = fz < end ? fz : end;                             // first non-zero in [beg,end), else end
		}
		const int nbeg = fz;
		if (lz < nbeg) lz = nbeg - 1;
		beg = nbeg;
		end = lz + 2 < qlen ? lz + 2 : qlen;
	}
	qle = best_j + 1; tle = best_i + 1; gtle = best_ie + 1; gscore_ = gscore; max_off_ = max_off;
	return best;
}

// ---------------------------------------------------------------------------------------------------
// ksw_global2 score only (ksw.c:504-584 without the backtrack matrix).  Same column layout.
// ---------------------------------------------------------------------------------------------------
template <int CPL>
__device__ int wave_global_score(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen, int w,
                                 Work &wk)
{
	const int l = lane(), j0 = l * CPL;
	const int oe_del = sw.o_del + sw.e_del, oe_ins = sw.o_ins + sw.e_ins, e_del = sw.e_del, e_ins = sw.e_ins;
	int qv[CPL], Hs[CPL], E[CPL];
#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int j = j0 + c;
		qv[c] = j < qlen ? q[j * qs] : 4;
		Hs[c] = j == 0 ? 0 : (j <= qlen && j <= w) ? -(sw.o_ins + e_ins * j) : NEG;   // ksw.c:523-526
		E[c] = NEG;
	}
	for (int i = 0; i < tlen; ++i) {
		const int tb = t[i * ts];
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int h1 = beg == 0 ? -(sw.o_del + e_del * (i + 1)) : NEG;
		wk.cells += end > beg ? (unsigned)(end - beg) : 0u;          // wavefront-uniform; lane 0 reports it
		if (CPL == 1) ++wk.rows1; else ++wk.rowsN;
		int M[CPL], u[CPL], P = LOW;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			M[c] = Hs[c] + sw.mat[tb * 5 + qv[c]];
			u[c] = inb ? M[c] - oe_ins + j * e_ins : LOW;
			P = P > u[c] ? P : u[c];
		}
		// F(i,beg) = MINUS_INF, F(i,j+1) = max(F(i,j) - e_ins, M_j - oe_ins): fold the initial value in as a
		// virtual column beg-1 carrying MINUS_INF + e_ins
		int run = wscan_excl_max(P, LOW);
		int h[CPL];
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			// F(i,j) = max( F_init - (j-beg)*e_ins , max_{beg<=k<j} u_k - (j-1)*e_ins ),  F_init = MINUS_INF
			int f = NEG - (j - beg) * e_ins;
			const int g = run - (j - 1) * e_ins;
			if (j > beg && run > LOW) f = f > g ? f : g;
			int hv = M[c] >= E[c] ? M[c] : E[c];
			hv = hv >= f ? hv : f;
			h[c] = hv;
			if (inb) {
				const int tD = M[c] - oe_del;
				int en = E[c] - e_del; en = en > tD ? en : tD;
				E[c] = en;
			}
			run = run > u[c] ? run : u[c];
		}
		const int up = __builtin_amdgcn_update_dpp(0, h[CPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
#pragma unroll
		for (int c = CPL - 1; c >= 0; --c) {
			const int j = j0 + c;
			const int left = c == 0 ? up : h[c - 1];
			if (j == beg) Hs[c] = h1;
			else if (j > beg && j <= end) Hs[c] = left;
			if (j == end) E[c] = NEG;                             // ksw.c:582
		}
		if (end == beg) {                                         // empty band: eh[end].h = h1 (only when qlen == 0; kept for fidelity)
#pragma unroll
			for (int c = 0; c < CPL; ++c) if (j0 + c == end) Hs[c] = h1;
		}
	}
	int score = LOW;
#pragma unroll
	for (int c = 0; c < CPL; ++c) if (j0 + c == qlen) score = Hs[c];
	return wmax(score);
}

// ---------------------------------------------------------------------------------------------------
// ksw_extend2 with a sliding column window.  The band of ksw_extend2 is trimmed after every row to the cells that are not zero
// (ksw.c:466-469), so the live columns [beg, end] of a row are far fewer than the flank is long -- about 40 at the start of an
// extension, growing with the score -- and they move down the diagonal.  wave_extend above gives every column of the flank a
// register for the whole extension (qlen / 64 columns per lane, paid in every row); here lane l owns columns base + l*CPL ..
// + CPL-1 of a window that starts at the band's first live column, with the fewest columns per lane (1..4) that hold the live
// band.  When the band runs out of the window, or has become narrow enough for fewer columns per lane, the H / E values of
// the live columns pass through LDS (two 16-bit values per column) and the rows go on in the instantiation that fits.  Each
// row is evaluated exactly as in wave_extend: same cells, same order of the row-level decisions.  One thing needs care: when the band
// grows by two columns in a row (ksw.c:469), the reference reads eh[] of a column no row has written since it was last inside the band
// -- zeros if it was trimmed away earlier (only all-zero cells are trimmed), the first-row value if no row ever reached it.  A column
// entering the window gets exactly that value (max_end = the rightmost column any row has written so far).
// ---------------------------------------------------------------------------------------------------
struct ExtSt { int i, beg, end, best, best_i, best_j, best_ie, gscore, max_off, max_end, hi; };   // hi: last column whose eh[] value sits in s_he
__device__ __forceinline__ void wsync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }   // one wavefront per workgroup: orders its LDS traffic
// eh[j].h of the first row (ksw.c:396-397); eh[j].e = 0
__device__ __forceinline__ int first_row_h(int j, int qlen, int h0, int oe_ins, int e_ins)
{
	const int h1st = h0 > oe_ins ? h0 - oe_ins : 0;
	if (j == 0) return h0;
	if (j == 1) return h1st;
	return (j <= qlen && h1st - (j - 2) * e_ins > e_ins) ? h1st - (j - 1) * e_ins : 0;
}
constexpr int WIN_MAX = 256;                                 // widest live band the windowed form takes (4 columns per lane)

template <int CPL>
__device__ __forceinline__ int ext_rows(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen,
                                     int w, int zdrop, int h0, ExtSt &S, unsigned *s_he, Work &wk)
{
	const int l = lane();
	qlen = __builtin_amdgcn_readfirstlane(qlen); tlen = __builtin_amdgcn_readfirstlane(tlen); w = __builtin_amdgcn_readfirstlane(w);
	zdrop = __builtin_amdgcn_readfirstlane(zdrop); h0 = __builtin_amdgcn_readfirstlane(h0);
	const int oe_del = __builtin_amdgcn_readfirstlane(sw.o_del + sw.e_del), oe_ins = __builtin_amdgcn_readfirstlane(sw.o_ins + sw.e_ins);
	const int e_del = __builtin_amdgcn_readfirstlane(sw.e_del), e_ins = __builtin_amdgcn_readfirstlane(sw.e_ins);
	int i = __builtin_amdgcn_readfirstlane(S.i), beg = __builtin_amdgcn_readfirstlane(S.beg), end = __builtin_amdgcn_readfirstlane(S.end);
	int best = __builtin_amdgcn_readfirstlane(S.best), best_i = __builtin_amdgcn_readfirstlane(S.best_i), best_j = __builtin_amdgcn_readfirstlane(S.best_j);
	int best_ie = __builtin_amdgcn_readfirstlane(S.best_ie), gscore = __builtin_amdgcn_readfirstlane(S.gscore), max_off = __builtin_amdgcn_readfirstlane(S.max_off);
	int max_end = __builtin_amdgcn_readfirstlane(S.max_end);
	const int hi = __builtin_amdgcn_readfirstlane(S.hi);
	const int base = beg;                                        // first column of the window: fixed for this run of rows
	const int j0 = base + l * CPL;
	int qv[CPL], Hs[CPL], E[CPL];
#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int j = j0 + c;
		qv[c] = j < qlen ? q[j * qs] : 4;
		// beyond `end`: what the reference's eh[] holds there -- zero where a row has been (trimmed cells are zero), else the first row
		const unsigned he = j <= hi ? s_he[j - base] : j > max_end ? (unsigned)first_row_h(j, qlen, h0, oe_ins, e_ins) : 0u;
		Hs[c] = (int)(he & 0xffffu); E[c] = (int)(he >> 16);
	}
	int scn[CPL];
	{
		const int tb0 = i < tlen ? t[i * ts] : 4;
#pragma unroll
		for (int c = 0; c < CPL; ++c) scn[c] = sw.mat[tb0 * 5 + qv[c]];
	}
	int status = 0;                                              // 0: the extension is over; 1: go on in another window
	for (; i < tlen; ++i) {
		int scv[CPL];
#pragma unroll
		for (int c = 0; c < CPL; ++c) scv[c] = scn[c];
		{
			const int tbn = i + 1 < tlen ? t[(i + 1) * ts] : 4;
#pragma unroll
			for (int c = 0; c < CPL; ++c) scn[c] = sw.mat[tbn * 5 + qv[c]];
		}
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		max_end = max_end > end ? max_end : end;
		int h1 = 0;
		if (beg == 0) { h1 = h0 - (sw.o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		wk.cells += end > beg ? (unsigned)(end - beg) : 0u;
		if (CPL == 1) ++wk.rows1; else ++wk.rowsN;
		int M[CPL], u[CPL], P = NEG;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			M[c] = Hs[c] ? Hs[c] + scv[c] : 0;
			int tI = M[c] - oe_ins; tI = tI > 0 ? tI : 0;
			u[c] = inb ? tI + j * e_ins : NEG;
			P = P > u[c] ? P : u[c];
		}
		int run = wscan_excl_max(P, NEG);
		int h[CPL], key = -1, firstnz = 1 << 30, lastnz = -1;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			const int f = j == beg ? 0 : run - (j - 1) * e_ins;
			int hv = M[c] > E[c] ? M[c] : E[c];
			hv = hv > f ? hv : f;
			h[c] = inb ? hv : 0;
			if (inb) {
				int tD = M[c] - oe_del; tD = tD > 0 ? tD : 0;
				int en = E[c] - e_del; en = en > tD ? en : tD;
				E[c] = en;
				const int k = hv * 1024 + j;
				key = key > k ? key : k;
			}
			run = run > u[c] ? run : u[c];
		}
		const int up = __builtin_amdgcn_update_dpp(0, h[CPL - 1], 0x138, 0xf, 0xf, false);   // wave_shr:1
#pragma unroll
		for (int c = CPL - 1; c >= 0; --c) {
			const int j = j0 + c;
			const int left = c == 0 ? up : h[c - 1];
			if (j == beg) Hs[c] = h1;
			else if (j > beg && j <= end) Hs[c] = left;
			if (j == end) E[c] = 0;
			if (j >= beg && j <= end && (Hs[c] != 0 || E[c] != 0)) { firstnz = firstnz < j ? firstnz : j; lastnz = lastnz > j ? lastnz : j; }
		}
		key = wmax(key);
		const int m = key < 0 ? 0 : key >> 10, mj = key < 0 ? -1 : key & 1023;
		if (end == qlen) {
			int hend = 0;
#pragma unroll
			for (int c = 0; c < CPL; ++c) if ((end - base) % CPL == c) hend = __builtin_amdgcn_readlane(Hs[c], (end - base) / CPL);
			if (end == beg) hend = h1;
			best_ie = gscore > hend ? best_ie : i;
			gscore = gscore > hend ? gscore : hend;
		}
		if (m == 0) break;
		if (m > best) {
			best = m; best_i = i; best_j = mj;
			const int off = mj > i ? mj - i : i - mj;
			max_off = max_off > off ? max_off : off;
		} else if (zdrop > 0) {
			if (i - best_i > mj - best_j) {
				if (best - m - ((i - best_i) - (mj - best_j)) * e_del > zdrop) break;
			} else {
				if (best - m - ((mj - best_j) - (i - best_i)) * e_ins > zdrop) break;
			}
		}
		const unsigned long long nzm = __ballot(lastnz >= 0);
		int fz = end, lz = -1;
		if (nzm) {
			fz = __builtin_amdgcn_readlane(firstnz, __ffsll((long long)nzm) - 1);
			lz = __builtin_amdgcn_readlane(lastnz, 63 - __clzll((long long)nzm));
			fz = fz < end ? fz : end;
		}
		const int nbeg = fz;
		if (lz < nbeg) lz = nbeg - 1;
		beg = nbeg;
		end = lz + 2 < qlen ? lz + 2 : qlen;
		// the window: column `end` needs a lane (it grows by one column per row at most: old_end was inside), and a band that has become
		// narrow enough for fewer columns per lane moves on to that instantiation (with slack, so that it does not come straight back)
		if (end - base > 64 * CPL - 1 || (CPL > 1 && end - beg + 1 <= 64 * (CPL - 1) - 16)) {
			if (i + 1 < tlen) {
				// a lane's registers hold eh[] of its columns as the reference's array would: live values up to this row's `end`, beyond it
				// what they were loaded with (see above); columns past the window's last lane are filled in by the next run's load
#pragma unroll
				for (int c = 0; c < CPL; ++c) {
					const int j = j0 + c;
					if (j >= beg && j <= end) s_he[j - beg] = ((unsigned)Hs[c] & 0xffffu) | (unsigned)E[c] << 16;
				}
				S.hi = end < base + 64 * CPL - 1 ? end : base + 64 * CPL - 1;
				status = 1; ++i;
				break;
			}
		}
	}
	S.max_end = max_end;
	S.i = i; S.beg = beg; S.end = end; S.best = best; S.best_i = best_i; S.best_j = best_j; S.best_ie = best_ie; S.gscore = gscore; S.max_off = max_off;
	return status;
}

// The driver: first row of ksw_extend2 (ksw.c:396-397), band clamp (ksw.c:399-407), then runs of rows in the window that fits.
// Flanks of fewer than 64 bases keep the plain one-column-per-lane form (nothing to gain there); so do bands wider than WIN_MAX
// columns (-w above 127) and scores that do not fit 16 bits.
template <int CPL>
__device__ __forceinline__ int wave_extend_fit(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen,
                                               int w, int end_bonus, int zdrop, int h0, int &qle, int &tle, int &gtle, int &gscore, int &max_off, Work &wk, unsigned *s_he)
{
	if (qlen < 64) return wave_extend<1>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
#ifdef KEXT_NO_WINDOW
	if (CPL > 2 && qlen < 128) return wave_extend<2>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
	if (CPL > 3 && qlen < 192) return wave_extend<3>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
	if (CPL > 4 && qlen < 256) return wave_extend<4>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
	return wave_extend<CPL>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
#endif
	const int e_ins = sw.e_ins, e_del = sw.e_del;
	int wc = w;                                                  // ksw.c:399-407
	{
		int max_ins = div_plus1_trunc(qlen * sw.mx + end_bonus - sw.o_ins, e_ins);
		max_ins = max_ins > 1 ? max_ins : 1;
		wc = wc < max_ins ? wc : max_ins;
		int max_del = div_plus1_trunc(qlen * sw.mx + end_bonus - sw.o_del, e_del);
		max_del = max_del > 1 ? max_del : 1;
		wc = wc < max_del ? wc : max_del;
	}
	const int end0 = qlen < wc + 1 ? qlen : wc + 1;              // row 0 reads columns [0, end0) and writes column end0
	if (end0 + 1 > WIN_MAX - 8 || h0 + qlen * sw.mx >= 32760)     // (-w above 127, or scores beyond 16 bits: the plain form, every column a register)
		return wave_extend<CPL>(sw, q, qs, qlen, t, ts, tlen, w, end_bonus, zdrop, h0, qle, tle, gtle, gscore, max_off, wk);
	ExtSt S;
	S.i = 0; S.beg = 0; S.end = end0; S.best = h0; S.best_i = -1; S.best_j = -1; S.best_ie = -1; S.gscore = -1; S.max_off = 0;
	S.max_end = -1; S.hi = -1;                                   // nothing in s_he yet: every column starts from the first row (ksw.c:396-397)
	while (S.i < tlen) {
		const int width = S.end - S.beg + 1;
		int st;
		if (width + 8 <= 64) st = ext_rows<1>(sw, q, qs, qlen, t, ts, tlen, wc, zdrop, h0, S, s_he, wk);
		else if (width + 8 <= 128) st = ext_rows<2>(sw, q, qs, qlen, t, ts, tlen, wc, zdrop, h0, S, s_he, wk);
		else if (CPL <= 3 || width + 8 <= 192) st = ext_rows<3>(sw, q, qs, qlen, t, ts, tlen, wc, zdrop, h0, S, s_he, wk);   // (CPL <= 3: flanks below 192 bases)
		else st = ext_rows<(CPL <= 3 ? 3 : 4)>(sw, q, qs, qlen, t, ts, tlen, wc, zdrop, h0, S, s_he, wk);
		wsync();
		if (!st) break;
	}
	qle = S.best_j + 1; tle = S.best_i + 1; gtle = S.best_ie + 1; gscore = S.gscore; max_off = S.max_off;
	return S.best;
}

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cal_max_gap(const DevOpt &o, int qlen)           // bwamem.c:628
{
	int l_del = div_plus1_trunc(qlen * o.a - o.o_del, o.e_del);
	int l_ins = div_plus1_trunc(qlen * o.a - o.o_ins, o.e_ins);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < o.w << 1 ? l : o.w << 1;
}

// Window of the reference any seed of the chain could reach (bwamem.c:649-664), clamped to the contig of the first
// seed as bns_fetch_seq does (bntseq.c:426), loaded into LDS as one base per byte.  false: window larger than MAXT.
__device__ __forceinline__ bool chain_window(const DevIndex &ix, const DevOpt &opt, const DevSeed *seeds, int n, int l_query,
                                             uint8_t *s_t, int t_cap, int l, int64_t &rmax0_out, int &tl_all_out)
{
	const int64_t l_pac = ix.l_pac;
	int64_t rmax0 = l_pac << 1, rmax1 = 0;
	for (int i = l; i < n; i += 64) {
		const DevSeed t = seeds[i];
		int64_t b = t.rbeg - (t.qbeg + cal_max_gap(opt, t.qbeg));
		int64_t e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + cal_max_gap(opt, l_query - t.qbeg - t.len));
		rmax0 = rmax0 < b ? rmax0 : b;
		rmax1 = rmax1 > e ? rmax1 : e;
	}
	rmax0 = wmin64(rmax0); rmax1 = wmax64(rmax1);
	rmax0 = rmax0 > 0 ? rmax0 : 0;
	rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
	const int64_t seed0_rbeg = seeds[0].rbeg;
	if (rmax0 < l_pac && l_pac < rmax1) {
		if (seed0_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
	}
	{
		const bool is_rev = seed0_rbeg >= l_pac;
		const int rid = dev_pos2rid(ix, is_rev ? (l_pac << 1) - 1 - seed0_rbeg : seed0_rbeg);
		int64_t far_beg = ix.anns[rid].offset, far_end = far_beg + ix.anns[rid].len;
		if (is_rev) { int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
		rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
		rmax1 = rmax1 < far_end ? rmax1 : far_end;
	}
	const int tl_all = (int)(rmax1 - rmax0);
	rmax0_out = rmax0; tl_all_out = tl_all;
	if (tl_all > t_cap) return false;
	__syncthreads();
	for (int i = l; i < tl_all; i += 64) s_t[i] = (uint8_t)ref_base(ix, rmax0 + i);
	return true;
}

// Extension of one seed into an alignment region (bwamem.c:716-793): left and right ksw_extend2 with band doubling
// (MAX_BAND_TRY = 2), clip-vs-to-end choice, seed coverage.  Needs s_q / s_t of the chain loaded.
template <int CPL>
__device__ __forceinline__ DevReg extend_seed(const Sw &sw, const DevOpt &opt, const uint8_t *s_q, const uint8_t *s_t, int l_query,
                                              int64_t rmax0, int tl_all, const DevSeed s, const DevChain &ch, const DevSeed *seeds, int n,
                                              int l, Work &wk, unsigned *s_he)
{
	DevReg reg;
	reg.rb = reg.re = 0; reg.frac_rep = 0; reg.qb = reg.qe = 0; reg.sub = reg.csub = reg.sub_n = 0; reg.seedcov = 0;
	reg.n_comp = 0; reg.is_alt = 0; reg.pad = 0;
	reg.rid = ch.rid; reg.score = reg.truesc = -1;
	int aw0 = opt.w, aw1 = opt.w;
	// left extension on the reversed prefixes (bwamem.c:721-750), then the right one (bwamem.c:752-780): one loop, so that the DP code
	// exists once per kernel
	for (int side = 0; side < 2; ++side) {
		const bool left = side == 0;
		const int qe = s.qbeg + s.len, re = (int)(s.rbeg + s.len - rmax0);
		if (left ? s.qbeg == 0 : qe == l_query) {
			if (left) { reg.score = reg.truesc = s.len * opt.a; reg.qb = 0; reg.rb = s.rbeg; }
			else { reg.qe = l_query; reg.re = s.rbeg + s.len; }
			continue;
		}
		const int tlen = left ? (int)(s.rbeg - rmax0) : tl_all - re;
		const int qlen = left ? s.qbeg : l_query - qe;
		const uint8_t *qp = left ? s_q + s.qbeg - 1 : s_q + qe;
		const uint8_t *tp = left ? s_t + tlen - 1 : s_t + re;
		const int st = left ? -1 : 1;
		const int bonus = left ? opt.pen_clip5 : opt.pen_clip3;
		const int sc0 = reg.score;                          // right side: the score the left side reached
		const int h0 = left ? s.len * opt.a : sc0;
		int qle = 0, tle = 0, gtle = 0, gscore = 0, max_off = 0, awc = opt.w;
		for (int i = 0; i < 2; ++i) {                       // MAX_BAND_TRY
			const int prev = reg.score;
			awc = opt.w << i;
			reg.score = wave_extend_fit<CPL>(sw, qp, st, qlen, tp, st, tlen, awc, bonus, opt.zdrop, h0, qle, tle, gtle, gscore, max_off, wk, s_he);
			if (reg.score == prev || max_off < (awc >> 1) + (awc >> 2)) break;
		}
		if (left) {
			aw0 = awc;
			if (gscore <= 0 || gscore <= reg.score - opt.pen_clip5) { reg.qb = s.qbeg - qle; reg.rb = s.rbeg - tle; reg.truesc = reg.score; }
			else { reg.qb = 0; reg.rb = s.rbeg - gtle; reg.truesc = gscore; }
		} else {
			aw1 = awc;
			if (gscore <= 0 || gscore <= reg.score - opt.pen_clip3) { reg.qe = qe + qle; reg.re = rmax0 + re + tle; reg.truesc += reg.score - sc0; }
			else { reg.qe = l_query; reg.re = rmax0 + re + gtle; reg.truesc += gscore - sc0; }
		}
	}
	int cov = 0;                                        // seedcov (bwamem.c:782-786)
	for (int i = l; i < n; i += 64) {
		const DevSeed t = seeds[i];
		if (t.qbeg >= reg.qb && t.qbeg + t.len <= reg.qe && t.rbeg >= reg.rb && t.rbeg + t.len <= reg.re) cov += t.len;
	}
	reg.seedcov = wsum(cov);
	reg.w = aw0 > aw1 ? aw0 : aw1;
	reg.seedlen0 = s.len;
	reg.frac_rep = ch.frac_rep;
	return reg;
}

constexpr int SPEC_NONE = -0x7fffffff - 1;                    // spec_regs[].score of a chain k_extend_spec could not take (window beyond LDS)

// K4a -- reads with many chains (hundreds, inside large repeat families) would keep one wavefront busy for tens of
// milliseconds while the rest of the GPU idles.  What k_extend must do in order is only the *decision* whether a seed
// is extended (it looks at the regions found so far); the extension itself depends on nothing but the seed and its
// chain.  So the best seed of every chain of such a read is extended here, one wavefront per chain, and k_extend
// picks the result up.  (A best seed that k_extend then skips was extended in vain; its result is never looked at.)
template <int CPL>
__global__ __launch_bounds__(64, (CPL <= 3 ? KEXT_W3 : CPL == 4 ? 4 : 1)) void k_extend_spec(ExtLaunch a)
{
	__shared__ uint8_t s_q[MAXQ + 8];
	__shared__ __attribute__((aligned(16))) uint8_t s_t[MAXT + 8];
	__shared__ int8_t s_mat[32];
	__shared__ __attribute__((aligned(16))) unsigned s_he_g[WIN_MAX + 128];   // reached through generic pointers too (the sorts' tables): 256 bytes in front and behind are never handed out
	unsigned *const s_he = s_he_g + 64;
	const int l = lane();
	const DevOpt &opt = a.opt;
	const DevIndex &ix = a.ix;
	Sw sw; sw.mat = s_mat; sw.o_del = opt.o_del; sw.e_del = opt.e_del; sw.o_ins = opt.o_ins; sw.e_ins = opt.e_ins;
	if (l < 25) s_mat[l] = opt.mat[l];
	sw.mx = wmax(l < 25 ? (int)opt.mat[l] : 0); if (sw.mx < 0) sw.mx = 0;
	Work wk = { 0, 0, 0 };
	const int n_items = *a.spec_n;
	for (int it = (int)blockIdx.x; it < n_items; it += (int)gridDim.x) {
		const int2 item = a.spec_items[it];
		const int r = item.x, ci = item.y;
		const int l_query = (int)(a.off[r + 1] - a.off[r]);
		const uint8_t *query = a.seq + a.off[r];
		const int64_t sb = a.seed_base[r];
		__syncthreads();
		for (int i = l; i < l_query; i += 64) s_q[i] = query[i];
		const DevChain ch = a.chains[sb + ci];
		const DevSeed *seeds = a.chain_seeds + sb + ch.seed_off;
		const int n = ch.n;
		if (n == 0) continue;
		int64_t rmax0; int tl_all;
		if (!chain_window(ix, opt, seeds, n, l_query, s_t, MAXT, l, rmax0, tl_all)) {          // k_extend hands the read to the large-window variant
			if (l == 0) a.spec_regs[sb + ci].score = SPEC_NONE;
			continue;
		}
		// the seed k_extend takes first: largest (score, index) (bwamem.c:669-674)
		long long best = -1;
		for (int i = l; i < n; i += 64) { const long long key = (long long)seeds[i].score << 32 | i; best = key > best ? key : best; }
		best = wmax64(best);
		const DevSeed s = seeds[(int)(best & 0xffffffff)];
		__syncthreads();
		const DevReg reg = extend_seed<CPL>(sw, opt, s_q, s_t, l_query, rmax0, tl_all, s, ch, seeds, n, l, wk, s_he);
		if (l == 0) a.spec_regs[sb + ci] = reg;
	}
	if (l == 0 && wk.cells) { atomicAdd(&cnt_row(a.counters)[CNT_CELLS], wk.cells); atomicAdd(&cnt_row(a.counters)[CNT_ROWS1], (unsigned long long)wk.rows1); atomicAdd(&cnt_row(a.counters)[CNT_ROWSN], (unsigned long long)wk.rowsN); }
}

// work list of k_extend_spec: (read, chain) for every chain of a read with at least min_chains chains
__global__ void k_spec_items(int n, const int *chain_n, int min_chains, int2 *items, int *n_items)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const int nc = chain_n[r];
	if (nc < min_chains) return;
	const int base = atomicAdd(n_items, nc);
	for (int ci = 0; ci < nc; ++ci) items[base + ci] = make_int2(r, ci);
}

// One read: all mem_chain2aln calls, then mem_sort_dedup_patch.  s_t holds the reference window of the current chain: LDS
// (T_CAP = MAXT) in k_extend; a read with a chain window or a patch window beyond that is handed over (redo list) to
// k_extend_big, the same code with the window in a global-memory slab of BIG_T bases (tandem repeats: a chain may drift
// by up to opt.w per merged seed, bwamem.c:203-217, so the window has no small bound; a wide -w widens it too).
constexpr int BIG_T = BWAHIP_EXT_BIG_T;
// mem_sort_dedup_patch (bwamem.c:444-496) for the ordinary short list -- at most FAST_N regions, no two keys equal in either sort, no pair of
// regions that mem_patch_reg would align -- in a lean kernel of its own (k_dedup_fast): the list lives in LDS, the sorts are counts of smaller keys by shuffle
// (without ties every correct sort is the reference's), the dedup pass runs on one lane.  Anything else (a tie, a patch candidate) returns -1
// with the global list untouched, and the read goes to k_dedup.  Returns the new length; the finished list is written back.
constexpr int FAST_N = 8;
__device__ __forceinline__ int fast_dedup(const DevOpt &opt, const DevIndex &ix, DevReg *gav, int n, DevReg *s_a, DevReg *s_b, int l)
{
	const int64_t l_pac = ix.l_pac;
	__syncthreads();
	// sort by re (ks_introsort(mem_ars2)): one region per lane
	int64_t key = 0;
	if (l < n) key = gav[l].re;
	int rank = 0; bool tie = false;
	for (int u = 0; u < n; ++u) {
		const int64_t ku = (int64_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)key, u) | (uint64_t)(uint32_t)__shfl((int)((uint64_t)key >> 32), u) << 32);
		rank += ku < key ? 1 : 0;
		tie |= ku == key && u != l;
	}
	if (__ballot(l < n && tie)) return -1;
	if (l < n) { DevReg v = gav[l]; v.n_comp = 1; s_a[rank] = v; }
	__threadfence_block(); __syncthreads();
	// the redundancy pass (bwamem.c:451-480) on lane 0; a pair mem_patch_reg would go on to align ends the fast path
	int bail = 0;
	if (l == 0) {
		for (int i = 1; i < n && !bail; ++i) {
			DevReg *p = &s_a[i];
			if (p->rid != s_a[i - 1].rid || p->rb >= s_a[i - 1].re + opt.max_chain_gap) continue;
			for (int j = i - 1; j >= 0 && p->rid == s_a[j].rid && p->rb < s_a[j].re + opt.max_chain_gap; --j) {
				DevReg *q = &s_a[j];
				if (q->qe == q->qb) continue;
				const int64_t orr = q->re - p->rb, oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
				const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
				const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
				if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
					if (p->score < q->score) { p->qe = p->qb; break; }
					else q->qe = q->qb;
				} else if (q->rb < p->rb) {                          // mem_patch_reg's tests before it aligns (bwamem.c:419-429)
					bool ok = true;
					if (q->rb < l_pac && p->rb >= l_pac) ok = false;
					if (ok && (q->qb >= p->qb || q->qe >= p->qe || q->re >= p->re)) ok = false;
					if (ok) {
						int w = (int)((q->re - p->rb) - (q->qe - p->qb));
						w = w > 0 ? w : -w;
						double rr = (double)(q->re - p->rb) / (p->re - q->rb) - (double)(q->qe - p->qb) / (p->qe - q->qb);
						rr = rr > 0. ? rr : -rr;
						if (q->re < p->rb || q->qe < p->qb) { if (w > opt.w << 1 || rr >= 0.05f) ok = false; }
						else if (w > opt.w << 2 || rr >= 0.05f * 2) ok = false;
					}
					if (ok) { bail = 1; break; }
				}
			}
		}
	}
	__threadfence_block(); __syncthreads();
	if (__shfl(bail, 0)) return -1;
	// drop the excluded entries (order kept), then sort by (score desc, rb, qb) (ks_introsort(mem_ars)); equal keys = identical hits: the full path's
	const bool keep = l < n && s_a[l].qe > s_a[l].qb;
	const unsigned long long km = __ballot(keep);
	const int m = __popcll(km);
	DevReg mine;
	if (keep) mine = s_a[l];
	const int pos = __popcll(km & ((1ull << l) - 1));               // its place after the compaction
	int sc = 0, qb = 0; int64_t rb = 0;
	if (keep) { sc = mine.score; rb = mine.rb; qb = mine.qb; }
	int rank2 = 0; bool tie2 = false;
	for (unsigned long long t = km; t; t &= t - 1) {
		const int u = __ffsll((long long)t) - 1;
		const int su = __shfl(sc, u), qu = __shfl(qb, u);
		const int64_t ru = (int64_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)rb, u) | (uint64_t)(uint32_t)__shfl((int)((uint64_t)rb >> 32), u) << 32);
		const bool lt_ut = su > sc || (su == sc && (ru < rb || (ru == rb && qu < qb)));
		rank2 += lt_ut ? 1 : 0;
		tie2 |= su == sc && ru == rb && qu == qb && u != l;
	}
	(void)pos;
	if (__ballot(keep && tie2)) return -1;
	if (keep) { if (mine.rid >= 0 && ix.anns[mine.rid].is_alt) mine.is_alt = 1; gav[rank2] = mine; }   // (bwamem.c:1091-1095 on the way out)
	__threadfence_block(); __syncthreads();
	return m;
}

// PHASE 0: both parts (k_extend_big); 1: the mem_chain2aln calls only -- a read left with more than one region is listed for k_dedup (its
// sorts and the dedup pass want other registers and run as their own launch); 2: mem_sort_dedup_patch of a listed read.
template <int CPL, bool BIGT, int PHASE>
__device__ __forceinline__ void extend_read(const ExtLaunch &a, const int r, uint8_t *s_q, uint8_t *s_t, int8_t *s_mat, int *s_stk, unsigned *s_he, unsigned *s_v = nullptr, int v_cap = 0)
{
	const int T_CAP = BIGT ? BIG_T : (a.lds_window < MAXT ? a.lds_window : MAXT);   // lds_window: test knob, forces the hand-over onto ordinary reads
	const int l = lane();
	const DevOpt &opt = a.opt;
	const DevIndex &ix = a.ix;
	const int l_query = (int)(a.off[r + 1] - a.off[r]);
	const uint8_t *query = a.seq + a.off[r];
	const int64_t sb = a.seed_base[r], rb0 = a.reg_base[r];
	const int n_chains = a.chain_n[r];
	const int64_t l_pac = ix.l_pac;
	const bool use_spec = !BIGT && a.spec_regs && n_chains >= a.spec_min_chains;
	DevReg *av = a.regs + rb0;                                  // the read's region list (av of bwamem.c:639)
	DevReg *const gav = av;
	int *srt = a.srt + 2 * sb;                                  // [0..n): seed index in ascending (score,idx) order; [n..2n): skipped flag
	int n_av = 0;
	Work wk = { 0, 0, 0 };
	const unsigned long long t_0 = wall_clock64();
	Sw sw; sw.mat = s_mat; sw.o_del = opt.o_del; sw.e_del = opt.e_del; sw.o_ins = opt.o_ins; sw.e_ins = opt.e_ins;
	__syncthreads();
	if (l < 25) s_mat[l] = opt.mat[l];
	sw.mx = wmax(l < 25 ? (int)opt.mat[l] : 0); if (sw.mx < 0) sw.mx = 0;
	bool have_q = PHASE != 2;                                   // the dedup pass needs the query only for mem_patch_reg's alignment (rare): fetched then
	if (have_q) for (int i = l; i < l_query; i += 64) s_q[i] = query[i];
	__syncthreads();

	if (PHASE == 2) n_av = a.reg_n[r];
	if (PHASE != 2)
	for (int ci = 0; ci < n_chains; ++ci) {
		const DevChain ch = a.chains[sb + ci];
		const DevSeed *seeds = a.chain_seeds + sb + ch.seed_off;
		const int n = ch.n;
		if (n == 0) continue;
		// the reference window of the chain: needed by extensions only.  A many-chain read got the best seed of every chain extended ahead of
		// time (k_extend_spec), and most of its chains have that one seed: the window (a contig look-up and a gather from the packed reference,
		// each a chain of dependent loads) is then fetched only if this kernel extends a seed of the chain itself
		int64_t rmax0 = 0; int tl_all = 0;
		bool have_win = false, win_fail = false;
		auto need_win = [&]() {
			if (have_win) return;
			have_win = true;
			if (!chain_window(ix, opt, seeds, n, l_query, s_t, T_CAP, l, rmax0, tl_all)) win_fail = true;
		};
		if (!use_spec) need_win();
		if (win_fail) {
			if (l == 0) {
				if (BIGT) { atomicExch(a.err, 3); atomicExch(a.err + 1, r); }          // window beyond BIG_T bases
				else a.redo_list[atomicAdd(a.redo_n, 1)] = r;                          // k_extend_big redoes the read
				a.reg_n[r] = 0;
			}
			return;
		}
		// ---- seeds in ascending (score<<32|index) order (bwamem.c:669-672; keys are unique, any sort does)
		for (int i = l; i < n; i += 64) {
			const int sc = seeds[i].score;
			int rank = 0;
			for (int u = 0; u < n; ++u) { const int su = seeds[u].score; rank += su < sc || (su == sc && u < i); }
			srt[rank] = i; srt[n + i] = 0;
		}
		__threadfence_block();
		__syncthreads();

		for (int k = n - 1; k >= 0; --k) {                      // best seed first (bwamem.c:674)
			const int sidx = srt[k];
			const DevSeed s = seeds[sidx];
			// ---- is the seed already covered by an earlier region? (bwamem.c:678-694)
			int hit = -1;
			for (int base = 0; base < n_av && hit < 0; base += 64) {
				const int i = base + l;
				bool brk = false;
				if (i < n_av) {
					const DevReg p = av[i];
					if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) &&
					    !(s.len - p.seedlen0 > .1 * l_query)) {
						int qd = s.qbeg - p.qb; int64_t rd = s.rbeg - p.rb;
						int max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
						int w = max_gap < p.w ? max_gap : p.w;
						if (qd - rd < w && rd - qd < w) brk = true;
						else {
							qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
							max_gap = cal_max_gap(opt, qd < rd ? qd : (int)rd);
							w = max_gap < p.w ? max_gap : p.w;
							if (qd - rd < w && rd - qd < w) brk = true;
						}
					}
				}
				const unsigned long long m = __ballot(brk);
				if (m) hit = base + __ffsll((long long)m) - 1;
			}
			if (hit >= 0) {                                     // bwamem.c:696-713: an overlapping, not-colinear better seed?
				bool any = false;
				for (int base = k + 1; base < n && !any; base += 64) {
					const int i = base + l;
					bool brk = false;
					if (i < n) {
						const int ti = srt[i];
						if (!srt[n + ti]) {                     // srt[i] != 0 (bwamem.c:701)
							const DevSeed t = seeds[ti];
							if (!(t.len < s.len * .95)) {
								if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) brk = true;
								if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) brk = true;
							}
						}
					}
					any = __ballot(brk) != 0;
				}
				if (!any) { if (l == 0) srt[n + sidx] = 1; __threadfence_block(); __syncthreads(); continue; }
			}
			// ---- extend (bwamem.c:716-793); the best seed of every chain of a many-chain read was extended ahead of
			// time by k_extend_spec (its result does not depend on the regions found so far, only the decision above does)
			DevReg reg;
			bool from_spec = use_spec && k == n - 1;
			if (from_spec) { reg = a.spec_regs[sb + ci]; if (reg.score == SPEC_NONE) from_spec = false; }
			if (!from_spec) {
				need_win();
				if (win_fail) {
					if (l == 0) { a.redo_list[atomicAdd(a.redo_n, 1)] = r; a.reg_n[r] = 0; }   // (use_spec implies !BIGT)
					return;
				}
				reg = extend_seed<CPL>(sw, opt, s_q, s_t, l_query, rmax0, tl_all, s, ch, seeds, n, l, wk, s_he);
			}
			if (l == 0) av[n_av] = reg;
			++n_av;
			__threadfence_block();
			__syncthreads();
		}
	}
	if (PHASE != 2 && a.dbg_regs) {                              // stage dump: regions before mem_sort_dedup_patch
		for (int i = l; i < n_av; i += 64) a.dbg_regs[rb0 + i] = av[i];
		if (l == 0) a.dbg_reg_n[r] = n_av;
	}

	const unsigned long long t_1 = wall_clock64();
	if (PHASE == 1 && n_av > 1) {                               // the list goes on to k_dedup
		if (l == 0) {
			a.reg_n[r] = n_av;
			const int li = a.subset == 1 ? 1 : 0;                   // the heavy reads' list is the second one
			a.dedup_list[(size_t)li * a.n_reads + atomicAdd(&a.dedup_n[li], 1)] = r;
			if (wk.cells) { atomicAdd(&cnt_row(a.counters)[CNT_CELLS], wk.cells); atomicAdd(&cnt_row(a.counters)[CNT_ROWS1], (unsigned long long)wk.rows1); atomicAdd(&cnt_row(a.counters)[CNT_ROWSN], (unsigned long long)wk.rowsN); }
			atomicMax(&cnt_row(a.counters)[14], t_1 - t_0);
		}
		return;
	}
	// ---- mem_sort_dedup_patch (bwamem.c:444-496) + is_alt (bwamem.c:1091-1095).  The scalar logic is replicated
	// in every lane (identical reads of av[]), stores are done by lane 0; the rare banded global alignment is
	// collective.  The sort permutes an index array; the list is then gathered into that order.
	int n = n_av;
	unsigned long long t_s1 = t_1, t_lp = t_1;
	// k_dedup, lists of up to 32 regions (nearly all): the list, its spare copy, the keys and the index array live in LDS for the whole
	// pass -- some fifteen dependent round trips per read otherwise go to global memory -- and the result is written back once
	constexpr int STAGE_N = 32;
	const bool stage = PHASE == 2 && s_v != nullptr && v_cap >= 2048 && n > 1 && n <= STAGE_N;
	DevReg *tmpv = a.tmp_regs + rb0;
	RegKey *keys = reinterpret_cast<RegKey*>(tmpv);             // (global form: the spare list is free whenever a sort runs)
	int *idx = srt;                                             // needs n ints (n <= number of seeds)
	int *work_lds = nullptr;
	if (stage) {
		uint8_t *L = reinterpret_cast<uint8_t*>(s_v);
		for (int i = l; i < n * 5; i += 64) reinterpret_cast<uint4*>(L)[i] = reinterpret_cast<const uint4*>(gav)[i];
		av = reinterpret_cast<DevReg*>(L); tmpv = reinterpret_cast<DevReg*>(L + STAGE_N * 80); keys = reinterpret_cast<RegKey*>(L + 2 * STAGE_N * 80);
		idx = reinterpret_cast<int*>(L + 2 * STAGE_N * 80 + STAGE_N * 16); work_lds = idx + 2 * STAGE_N;
		__threadfence_block(); __syncthreads();
	}
	if (n > 1) {
		// sort by re
		for (int i = l; i < n; i += 64) { keys[i].k64 = av[i].re; keys[i].score = 0; keys[i].qb = 0; idx[i] = i; }
		__threadfence_block(); __syncthreads();
		// ks_introsort(mem_ars2): by the whole wavefront, exact also with equal keys (regsort_dev.h / isort_dev.h); the one-lane restatement
		// only when the introsort's depth limit is reached (or the test knob asks for it).  Scratch: behind the keys in the spare list.
		if (n < a.rank_sort_min || !wave_sort_exact(RegSort{keys, 0}, n, idx, stage ? work_lds : reinterpret_cast<int*>(keys + n), s_stk, s_he, l, stage ? nullptr : s_v, stage ? 0 : v_cap)) {
			__threadfence_block(); __syncthreads();
			if (l == 0) { int bad = 0; rs_introsort(RegSort{keys, 0}, n, idx, s_stk, &bad); if (bad) atomicExch(a.err, 10 + bad); }
		}
		__threadfence_block(); __syncthreads();
		t_s1 = wall_clock64();
		// gather into sorted order through the spare list (all lanes), then copy back
		for (int i = l; i < n; i += 64) tmpv[i] = av[idx[i]];
		__threadfence_block(); __syncthreads();
		for (int i = l; i < n; i += 64) av[i] = tmpv[i];
		__threadfence_block(); __syncthreads();
		if (l == 0) for (int i = 0; i < n; ++i) av[i].n_comp = 1;
		__threadfence_block(); __syncthreads();
		// bwamem.c:453: an entry takes part only if it lies within max_chain_gap of its predecessor on the same contig.  That test reads rid and
		// re, which this loop never changes, and the entry's own rb, which only its own turn changes: it is evaluated for 64 entries at a
		// time ahead of the sequential pass, which then visits the entries that passed (a read inside a repeat family has hundreds of
		// regions at unrelated places: nearly all of them are skipped)
		for (int cbase = 0; cbase < n; cbase += 64) {
		unsigned long long act_m;
		{
			const int ii = cbase + l;
			bool act = false;
			if (ii >= 1 && ii < n) act = av[ii].rid == av[ii - 1].rid && av[ii].rb < av[ii - 1].re + opt.max_chain_gap;
			act_m = __ballot(act);
		}
		while (act_m) {
			const int i = cbase + __ffsll((long long)act_m) - 1;
			act_m &= act_m - 1;
			DevReg p = av[i];
			bool p_dirty = false;
			for (int j = i - 1; j >= 0; --j) {
				DevReg q = av[j];
				if (!(p.rid == q.rid && p.rb < q.re + opt.max_chain_gap)) break;
				if (q.qe == q.qb) continue;                     // excluded
				const int64_t orr = q.re - p.rb;
				const int64_t oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				const int64_t mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
				const int64_t mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				if ((float)orr > opt.mask_level_redun * (float)mr && (float)oq > opt.mask_level_redun * (float)mq) {
					if (p.score < q.score) { p.qe = p.qb; p_dirty = true; break; }
					else { q.qe = q.qb; if (l == 0) av[j].qe = q.qe; __threadfence_block(); __syncthreads(); }
				} else if (q.rb < p.rb) {
					// ---- mem_patch_reg (bwamem.c:413-442), a = q, b = p
					int score = 0, wband = 0;
					bool ok = true;
					if (q.rb < l_pac && p.rb >= l_pac) ok = false;
					if (ok && (q.qb >= p.qb || q.qe >= p.qe || q.re >= p.re)) ok = false;
					if (ok) {
						int w = (int)((q.re - p.rb) - (q.qe - p.qb));
						w = w > 0 ? w : -w;
						double rr = (double)(q.re - p.rb) / (p.re - q.rb) - (double)(q.qe - p.qb) / (p.qe - q.qb);
						rr = rr > 0. ? rr : -rr;
						if (q.re < p.rb || q.qe < p.qb) { if (w > opt.w << 1 || rr >= 0.05f) ok = false; }
						else if (w > opt.w << 2 || rr >= 0.05f * 2) ok = false;
						if (ok) {
							w += q.w + p.w;
							w = w < opt.w << 2 ? w : opt.w << 2;
							// bwa_gen_cigar2 score only (bwa.c:261-307): query[q.qb, p.qe) vs ref [q.rb, p.re)
							const int lq = p.qe - q.qb;
							const int64_t grb = q.rb, gre = p.re;
							int gsc = 0;
							bool have = !(lq <= 0 || grb >= gre || (grb < l_pac && gre > l_pac));
							const int rlen = (int)(gre - grb);
								if (have && rlen > T_CAP) {                      // patch window beyond the LDS window: same hand-over
									if (l == 0) {
										if (BIGT) { atomicExch(a.err, 4); atomicExch(a.err + 1, r); }
										else a.redo_list[atomicAdd(a.redo_n, 1)] = r;
										a.reg_n[r] = 0;
									}
									return;
								}
							if (have) {
								__syncthreads();
								if (!have_q) { for (int t = l; t < l_query; t += 64) s_q[t] = query[t]; have_q = true; }
								for (int t = l; t < rlen; t += 64) s_t[t] = (uint8_t)ref_base(ix, grb + t);
								__syncthreads();
								const bool rev = grb >= l_pac;      // both reversed so gaps are left-aligned on the forward strand (bwa.c:275-280)
								if (lq == rlen && w == 0) {
									int part = 0;
									for (int t = l; t < lq; t += 64) part += s_mat[s_t[t] * 5 + s_q[q.qb + t]];
									gsc = wsum(part);
								} else {
									int max_ins = div_plus1_trunc(((lq + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins);
									int max_del = div_plus1_trunc(((lq + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del);
									int max_gap = max_ins > max_del ? max_ins : max_del;
									max_gap = max_gap > 1 ? max_gap : 1;
									int dl = rlen - lq; dl = dl < 0 ? -dl : dl;
									int gw = (max_gap + dl + 1) >> 1;
									gw = gw < w ? gw : w;
									const int min_w = dl + 3;
									gw = gw > min_w ? gw : min_w;
									gsc = rev ? wave_global_score<CPL>(sw, s_q + p.qe - 1, -1, lq, s_t + rlen - 1, -1, rlen, gw, wk)
									          : wave_global_score<CPL>(sw, s_q + q.qb, 1, lq, s_t, 1, rlen, gw, wk);
								}
								score = gsc;
								const int q_s = (int)((double)(p.qe - q.qb) / ((p.qe - p.qb) + (q.qe - q.qb)) * (p.score + q.score) + .499);
								const int r_s = (int)((double)(p.re - q.rb) / ((p.re - p.rb) + (q.re - q.rb)) * (p.score + q.score) + .499);
								if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) score = 0;
								wband = w;
							} else score = 0;                   // bwa_gen_cigar2 returned without a score: the reference reads an
							                                     // uninitialised int here; unreachable for regions of one chain strand
						} else score = 0;
					}
					if (ok && score > 0) {                      // merge q into p (bwamem.c:470-478)
						p.n_comp += q.n_comp + 1;
						p.seedcov = p.seedcov > q.seedcov ? p.seedcov : q.seedcov;
						p.sub = p.sub > q.sub ? p.sub : q.sub;
						p.csub = p.csub > q.csub ? p.csub : q.csub;
						p.qb = q.qb; p.rb = q.rb;
						p.truesc = p.score = score;
						p.w = wband;
						q.qb = q.qe;
						p_dirty = true;
						if (l == 0) av[j].qb = q.qb;
						__threadfence_block(); __syncthreads();
					}
				}
			}
			if (p_dirty) {                                      // write back only what a merge / exclusion can change
				if (l == 0) {
					DevReg *d = &av[i];
					d->qe = p.qe; d->qb = p.qb; d->rb = p.rb; d->n_comp = p.n_comp; d->seedcov = p.seedcov;
					d->sub = p.sub; d->csub = p.csub; d->truesc = p.truesc; d->score = p.score; d->w = p.w;
				}
				__threadfence_block(); __syncthreads();
			}
		}
		}
		t_lp = wall_clock64();
		// compact, sort by (score desc, rb, qb), drop identical hits (bwamem.c:481-495)
		{                                                       // drop the excluded entries (qe == qb), order kept: ballot compaction
			int m = 0;
			for (int base = 0; base < n; base += 64) {
				const int i = base + l;
				const bool keep = i < n && av[i].qe > av[i].qb;
				const unsigned long long km = __ballot(keep);
				if (keep) tmpv[m + __popcll(km & ((1ull << l) - 1))] = av[i];
				m += __popcll(km);
			}
			__threadfence_block(); __syncthreads();
			n = m;
			for (int i = l; i < n; i += 64) av[i] = tmpv[i];
			__threadfence_block(); __syncthreads();
		}
		for (int i = l; i < n; i += 64) { keys[i].k64 = av[i].rb; keys[i].score = av[i].score; keys[i].qb = av[i].qb; idx[i] = i; }
		__threadfence_block(); __syncthreads();
		if (n < a.rank_sort_min || !wave_sort_exact(RegSort{keys, 1}, n, idx, stage ? work_lds : reinterpret_cast<int*>(keys + n), s_stk, s_he, l, stage ? nullptr : s_v, stage ? 0 : v_cap)) {
			__threadfence_block(); __syncthreads();
			if (l == 0) { int bad = 0; rs_introsort(RegSort{keys, 1}, n, idx, s_stk, &bad); if (bad) atomicExch(a.err, 20 + bad); }
		}
		__threadfence_block(); __syncthreads();
		for (int i = l; i < n; i += 64) tmpv[i] = av[idx[i]];
		__threadfence_block(); __syncthreads();
		for (int i = l; i < n; i += 64) av[i] = tmpv[i];
		__threadfence_block(); __syncthreads();
		{                                                       // identical hits (same score, rb, qb as the predecessor) go; bwamem.c:488-494
			int m = 0;
			for (int base = 0; base < n; base += 64) {
				const int i = base + l;
				bool keep = i < n;
				if (i > 0 && i < n) keep = !(av[i].score == av[i-1].score && av[i].rb == av[i-1].rb && av[i].qb == av[i-1].qb);
				const unsigned long long km = __ballot(keep);
				if (keep) tmpv[m + __popcll(km & ((1ull << l) - 1))] = av[i];
				m += __popcll(km);
			}
			__threadfence_block(); __syncthreads();
			n = m;
			for (int i = l; i < n; i += 64) av[i] = tmpv[i];
			__threadfence_block(); __syncthreads();
		}
	}
	for (int i = l; i < n; i += 64) {                           // bwamem.c:1091-1095
		if (av[i].rid >= 0 && ix.anns[av[i].rid].is_alt) av[i].is_alt = 1;
	}
	if (stage) {                                                // the finished list back to its global slots
		__threadfence_block(); __syncthreads();
		for (int i = l; i < n * 5; i += 64) reinterpret_cast<uint4*>(gav)[i] = reinterpret_cast<const uint4*>(av)[i];
	}
	if (l == 0) {
		a.reg_n[r] = n;
		if (wk.cells) { atomicAdd(&cnt_row(a.counters)[CNT_CELLS], wk.cells); atomicAdd(&cnt_row(a.counters)[CNT_ROWS1], (unsigned long long)wk.rows1); atomicAdd(&cnt_row(a.counters)[CNT_ROWSN], (unsigned long long)wk.rowsN); }
		if (PHASE != 2) atomicMax(&cnt_row(a.counters)[14], t_1 - t_0);
		atomicMax(&cnt_row(a.counters)[15], wall_clock64() - t_1);
		atomicMax(&cnt_row(a.counters)[21], t_s1 - t_1); atomicMax(&cnt_row(a.counters)[22], t_lp - t_s1); atomicMax(&cnt_row(a.counters)[23], wall_clock64() - t_lp);
	}
}

template <int CPL>
// (CPL <= 3: seven waves per SIMD at 72 VGPRs and 128 bytes of spill beat five at 96 VGPRs by 6 % -- measured back to back on one box)
__global__ __launch_bounds__(64, (CPL <= 3 ? KEXT_W3 : CPL == 4 ? 4 : 1)) void k_extend(ExtLaunch a)
{
	__shared__ uint8_t s_q[MAXQ + 8];
	__shared__ __attribute__((aligned(16))) uint8_t s_t[MAXT + 8];
	__shared__ int8_t s_mat[32];
	__shared__ int s_stk_g[3 * 80 + 32];
	int *const s_stk = s_stk_g + 16;
	__shared__ __attribute__((aligned(16))) unsigned s_he_g[WIN_MAX + 128];   // reached through generic pointers too (the sorts' tables): 256 bytes in front and behind are never handed out
	unsigned *const s_he = s_he_g + 64;
	// a.subset 1: the reads with many seeds left (the first two classes of the launch order), on their own stream with their own k_dedup behind
	// them, so that their long extensions AND their long lists run beside the bulk; 2: all other reads; 0: everything (no launch order)
	const int n_heavy = a.perm ? a.perm_counts[0] + a.perm_counts[1] : 0;
	for (int idx = (int)blockIdx.x;; idx += (int)gridDim.x) {     // (one call site: subset 1 strides over the heavy reads, the others take one read)
		if (a.subset == 1 ? idx >= n_heavy : (a.subset == 2 && idx < n_heavy)) break;
		extend_read<CPL, false, 1>(a, a.perm ? a.perm[idx] : idx, s_q, s_t, s_mat, s_stk, s_he);
		if (a.subset != 1) break;
		__syncthreads();
	}
}

// The bulk's lists first go through fast_dedup, one read per wavefront: a quarter of all reads has two or three regions, and the full
// k_dedup spends some fifteen dependent round trips on each.  What fast_dedup does not take goes on to list 2 for k_dedup.
__global__ __launch_bounds__(64) void k_dedup_fast(ExtLaunch a)
{
	__shared__ __attribute__((aligned(16))) DevReg s_g[2 * FAST_N + 8];   // (4 unused records at either end: the lists are reached through generic pointers)
	const int l = lane();
	const int n_list = a.dedup_n[0];
	for (int it = (int)blockIdx.x; it < n_list; it += (int)gridDim.x) {
		const int r = a.dedup_list[it];
		const int n = a.reg_n[r];
		int m = -1;
		if (n <= FAST_N) m = fast_dedup(a.opt, a.ix, a.regs + a.reg_base[r], n, s_g + 4, s_g + 4 + FAST_N, l);
		if (l == 0) {
			if (m >= 0) a.reg_n[r] = m;
			else a.dedup_list[(size_t)2 * a.n_reads + atomicAdd(&a.dedup_n[2], 1)] = r;
		}
		__syncthreads();
	}
}

// mem_sort_dedup_patch of the reads k_extend listed (more than one region), one read per wavefront
template <int CPL>
__global__ __launch_bounds__(64) void k_dedup(ExtLaunch a)
{
	__shared__ uint8_t s_q[MAXQ + 8];
	__shared__ __attribute__((aligned(16))) uint8_t s_t[MAXT + 8];
	__shared__ int8_t s_mat[32];
	__shared__ int s_stk_g[3 * 80 + 32];
	int *const s_stk = s_stk_g + 16;
	__shared__ __attribute__((aligned(16))) unsigned s_he_g[WIN_MAX + 128];   // reached through generic pointers too (the sorts' tables): 256 bytes in front and behind are never handed out
	unsigned *const s_he = s_he_g + 64;
	// the sort's working array for lists up to 2048 regions (longer ones: global memory); short lists live here whole.  It is reached through
	// generic pointers, so 256 bytes in front and behind are never handed out: whatever constant offset the compiler folds into a flat
	// instruction, its base register stays inside the LDS aperture (the array may well sit at LDS offset 0)
	__shared__ __attribute__((aligned(16))) unsigned s_v_g[2048 + 128];
	unsigned *const s_v = s_v_g + 64;
	const int li = a.subset == 1 ? 1 : a.subset == 3 ? 2 : 0;   // the heavy reads' list, the list k_dedup_fast passed on, or all of the bulk
	const int n_list = a.dedup_n[li];
	for (int it = (int)blockIdx.x; it < n_list; it += (int)gridDim.x) {
		extend_read<CPL, false, 2>(a, a.dedup_list[(size_t)li * a.n_reads + n_list - 1 - it], s_q, s_t, s_mat, s_stk, s_he, s_v, 2048);   // from the end of the list: the reads k_extend finished last are the ones with the longest lists
		__syncthreads();
	}
}

// the reads k_extend handed over (reference window beyond LDS): window in this workgroup's global slab
template <int CPL>
__global__ __launch_bounds__(64) void k_extend_big(ExtLaunch a)
{
	__shared__ uint8_t s_q[MAXQ + 8];
	__shared__ int8_t s_mat[32];
	__shared__ int s_stk_g[3 * 80 + 32];
	int *const s_stk = s_stk_g + 16;
	__shared__ __attribute__((aligned(16))) unsigned s_he_g[WIN_MAX + 128];   // reached through generic pointers too (the sorts' tables): 256 bytes in front and behind are never handed out
	unsigned *const s_he = s_he_g + 64;
	uint8_t *s_t = a.big_t + (size_t)blockIdx.x * (BIG_T + 64);
	const int n_redo = *a.redo_n;
	for (int it = (int)blockIdx.x; it < n_redo; it += (int)gridDim.x) {
		extend_read<CPL, true, 0>(a, a.redo_list[it], s_q, s_t, s_mat, s_stk, s_he);
		__syncthreads();
	}
}

// Scheduling aid: reads with many seeds to extend go to the front of the launch order, the heaviest first, so that the long ones
// start at once and the short ones fill in behind them (the order has no effect on results).  Four classes by the number of seeds
// left after chain filtering; counts[0..3]: class sizes, counts[4..7]: cursors.
__device__ __forceinline__ int order_class(int v, int t0, int t1, int t2) { return v >= t0 ? 0 : v >= t1 ? 1 : v >= t2 ? 2 : 3; }
__global__ void k_order_count(int n, const int *kept_seeds, int t0, int t1, int t2, int *counts)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	const int c = r < n ? order_class(kept_seeds[r], t0, t1, t2) : 3;
	// one atomic per class and wavefront
	for (int k = 0; k < 4; ++k) {
		const unsigned long long m = __ballot(r < n && c == k);
		if (m && (threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&counts[k], __popcll(m));
	}
}
__global__ void k_order_place(int n, const int *kept_seeds, int t0, int t1, int t2, int *perm, int *counts)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	const int c = r < n ? order_class(kept_seeds[r], t0, t1, t2) : 3;
	const int l = threadIdx.x & 63;
	for (int k = 0; k < 4; ++k) {
		const unsigned long long m = __ballot(r < n && c == k);
		if (!m) continue;
		int base = 0;
		const int lead = __ffsll((long long)m) - 1;
		if (l == lead) base = atomicAdd(&counts[4 + k], __popcll(m));
		base = __shfl(base, lead);
		int start = 0;
		for (int j = 0; j < k; ++j) start += counts[j];
		if (r < n && c == k) perm[start + base + __popcll(m & ((1ull << l) - 1))] = r;
	}
}

// known-answer kernel: ksw_extend2 on caller-supplied pairs (params per item: qlen,tlen,w,h0,zdrop,end_bonus,o_del,e_del,o_ins,e_ins)
__global__ __launch_bounds__(64) void k_kat_ksw(DevOpt opt, int n, const int *params, const uint8_t *q, const int64_t *qoff,
                                                const uint8_t *t, const int64_t *toff, int *out6)
{
	__shared__ uint8_t s_q[MAXQ + 8];
	__shared__ __attribute__((aligned(16))) uint8_t s_t[MAXT + 8];
	__shared__ int8_t s_mat[32];
	__shared__ __attribute__((aligned(16))) unsigned s_he_g[WIN_MAX + 128];   // reached through generic pointers too (the sorts' tables): 256 bytes in front and behind are never handed out
	unsigned *const s_he = s_he_g + 64;
	const int r = blockIdx.x, l = lane();
	if (r >= n) return;
	const int *p = params + 10 * r;
	const int qlen = p[0], tlen = p[1];
	if (l < 25) s_mat[l] = opt.mat[l];
	for (int i = l; i < qlen; i += 64) s_q[i] = q[qoff[r] + i];
	for (int i = l; i < tlen; i += 64) s_t[i] = t[toff[r] + i];
	__syncthreads();
	Sw sw; sw.mat = s_mat; sw.o_del = p[6]; sw.e_del = p[7]; sw.o_ins = p[8]; sw.e_ins = p[9];
	sw.mx = wmax(l < 25 ? (int)opt.mat[l] : 0); if (sw.mx < 0) sw.mx = 0;
	int qle, tle, gtle, gscore, max_off;
	Work wk = { 0, 0, 0 };
	int sc = wave_extend<11>(sw, s_q, 1, qlen, s_t, 1, tlen, p[2], p[5], p[4], p[3], qle, tle, gtle, gscore, max_off, wk);
	// the windowed form (what the extension kernels run for flanks of 64 bases and more) must give the same six numbers
	int qle2, tle2, gtle2, gscore2, max_off2;
	const int sc2 = wave_extend_fit<11>(sw, s_q, 1, qlen, s_t, 1, tlen, p[2], p[5], p[4], p[3], qle2, tle2, gtle2, gscore2, max_off2, wk, s_he);
	if (sc2 != sc || qle2 != qle || tle2 != tle || gtle2 != gtle || gscore2 != gscore || max_off2 != max_off) sc = -777777;
	if (l == 0) { int *o = out6 + 6 * r; o[0] = sc; o[1] = qle; o[2] = tle; o[3] = gtle; o[4] = gscore; o[5] = max_off; }
}

// known-answer kernel: the wavefront's exact introsort (regsort_dev.h / isort_dev.h) beside the one-lane restatement of ksort.h on the same keys
__global__ __launch_bounds__(64) void k_kat_isort(int n, int mode, const RegKey *keys, int *idx_par, int *idx_seq, int *work, int *status)
{
	__shared__ int s_stk_g[3 * 80 + 32];
	int *const s_stk = s_stk_g + 16;
	__shared__ __attribute__((aligned(16))) unsigned s_lds_g[256 + 128];
	unsigned *const s_lds = s_lds_g + 64;
	const int l = lane();
	for (int i = l; i < n; i += 64) { idx_par[i] = i; idx_seq[i] = i; }
	__threadfence_block(); __syncthreads();
	const bool ok = wave_sort_exact(RegSort{keys, mode}, n, idx_par, work, s_stk, s_lds, l);
	__threadfence_block(); __syncthreads();
	if (l == 0) { int bad = 0; rs_introsort(RegSort{keys, mode}, n, idx_seq, s_stk, &bad); status[0] = ok ? 1 : 0; status[1] = bad; }
}

} // namespace

int launch_kat_isort(int n, int mode, const void *keys16, int *idx_par, int *idx_seq, int *work, int *status, hipStream_t st)
{
	hipLaunchKernelGGL(k_kat_isort, dim3(1), dim3(64), 0, st, n, mode, reinterpret_cast<const RegKey*>(keys16), idx_par, idx_seq, work, status);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
size_t kat_isort_work_ints(int n) { return 6 * ((size_t)n / 64 + 1) + 2 * ((size_t)n / 2 + 1) + (size_t)n + (size_t)n / 4 + 4; }

int launch_kat_ksw(const DevOpt &opt, int n, const int *params, const uint8_t *q, const int64_t *qoff, const uint8_t *t, const int64_t *toff,
                   int *out6, hipStream_t st)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_kat_ksw, dim3(n), dim3(64), 0, st, opt, n, params, q, qoff, t, toff, out6);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

// perm[0..n): the reads ordered by class of keys[r] (>= t0, >= t1, >= t2, rest); counts: 8 ints of scratch
int launch_order(int n, const int *keys, int t0, int t1, int t2, int *perm, int *counts, hipStream_t st)
{
	if (n <= 0) return 0;
	(void)hipMemsetAsync(counts, 0, 32, st);
	hipLaunchKernelGGL(k_order_count, dim3((n + 255) / 256), dim3(256), 0, st, n, keys, t0, t1, t2, counts);
	hipLaunchKernelGGL(k_order_place, dim3((n + 255) / 256), dim3(256), 0, st, n, keys, t0, t1, t2, perm, counts);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_extend_spec(const ExtLaunch &a, int max_len, hipStream_t st)
{
	if (a.n_reads <= 0 || !a.spec_regs) return 0;
	(void)hipMemsetAsync(a.spec_n, 0, 4, st);
	hipLaunchKernelGGL(k_spec_items, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, a.n_reads, a.chain_n, a.spec_min_chains, a.spec_items, a.spec_n);
	const int grid = 16384;
	if (max_len + 1 <= 64 * 3) hipLaunchKernelGGL(k_extend_spec<3>, dim3(grid), dim3(64), 0, st, a);
	else if (max_len + 1 <= 64 * 4) hipLaunchKernelGGL(k_extend_spec<4>, dim3(grid), dim3(64), 0, st, a);
	else if (max_len + 1 <= 64 * 5) hipLaunchKernelGGL(k_extend_spec<5>, dim3(grid), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_extend_spec<11>, dim3(grid), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

template <int CPL> static void launch_extend_cpl(ExtLaunch a, hipStream_t st, hipStream_t st2)
{
	const int dgrid = a.n_reads < 32768 ? a.n_reads : 32768;
	if (a.perm && st2) {
		a.subset = 1;                                            // the heavy reads: extension, then their lists, on the second stream
		hipLaunchKernelGGL(k_extend<CPL>, dim3(a.n_reads < 4096 ? a.n_reads : 4096), dim3(64), 0, st2, a);
		hipLaunchKernelGGL(k_dedup<CPL>, dim3(a.n_reads < 4096 ? a.n_reads : 4096), dim3(64), 0, st2, a);
		a.subset = 2;
	} else a.subset = 0;
	hipLaunchKernelGGL(k_extend<CPL>, dim3(a.n_reads), dim3(64), 0, st, a);
	hipLaunchKernelGGL(k_dedup_fast, dim3(dgrid), dim3(64), 0, st, a);   // the ordinary short lists; the rest goes on to list 2
	a.subset = 3;
	hipLaunchKernelGGL(k_dedup<CPL>, dim3(dgrid), dim3(64), 0, st, a);
}

// st2 / fork / join: a second stream (and two events) of the context; nullptr = everything on st
int launch_extend(const ExtLaunch &a, int max_len, hipStream_t st, hipStream_t st2, hipEvent_t fork, hipEvent_t join)
{
	if (a.n_reads <= 0) return 0;
	if (a.perm) launch_order(a.n_reads, a.kept_seeds, 1024, 256, 64, a.perm, a.perm_counts, st);
	const bool two = a.perm && st2 && fork && join;
	if (two && (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(st2, fork, 0) != hipSuccess)) return BWAHIP_ENODEV;
	// columns 0..max_len must fit in 64 lanes x CPL registers
	if (max_len + 1 <= 64 * 3) launch_extend_cpl<3>(a, st, two ? st2 : nullptr);
	else if (max_len + 1 <= 64 * 4) launch_extend_cpl<4>(a, st, two ? st2 : nullptr);
	else if (max_len + 1 <= 64 * 5) launch_extend_cpl<5>(a, st, two ? st2 : nullptr);
	else launch_extend_cpl<11>(a, st, two ? st2 : nullptr);
	if (two && (hipEventRecord(join, st2) != hipSuccess || hipStreamWaitEvent(st, join, 0) != hipSuccess)) return BWAHIP_ENODEV;
	// reads whose reference window exceeded the LDS window (none on ordinary data): one generic instantiation
	hipLaunchKernelGGL(k_extend_big<11>, dim3(BWAHIP_EXT_BIG_GRID), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
