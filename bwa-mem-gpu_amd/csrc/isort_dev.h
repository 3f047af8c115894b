// klib's ks_introsort (ksort.h:176-227), reproduced exactly by a whole wavefront.
//
// The reference sorts region / chain lists with an UNSTABLE introsort, so the order of elements with equal keys is a property of
// that very algorithm, and it is visible in the output (which of two duplicate hits survives mem_sort_dedup_patch, bwamem.c:460-478).
// When no two keys are equal any correct sort gives the same permutation (rank sort, regsort_dev.h).  With ties the algorithm
// itself has to be followed -- and run on one lane it is a chain of dependent memory round trips (n log n of them) that sets the
// duration of a kernel whenever a read has hundreds of regions.  Two facts make it parallel without changing its result:
//
//  1. The partition step (ksort.h:199-205: `do ++i while (a[i] < p); do --j while (i <= j && p < a[j]); swap`) is a function of
//     two flag vectors.  With L = positions in (s, t] where the upward scan stops (!(a[x] < p)), ascending, and R = positions in
//     [s, t) where the downward scan stops (!(p < a[y])), descending, the loop swaps l_k <-> r_k for k = 1..K, K = the number of
//     k with l_k < r_k, and ends with i = min(l_{K+1}, r_K) (i = l_1 when K = 0).  Flags are ballots, ranks are popcounts, the
//     swaps are independent: one pass over the range with 64 lanes instead of one step per element.
//  2. The final __ks_insertsort over the whole array moves an element left only past strictly greater ones: it is a stable sort of
//     the arrangement the quicksort phase leaves.  So the final slot of an element is (number of smaller keys) + (number of equal
//     keys before it in that arrangement), both counted in parallel.
//
// Elements are handled as packed words (number of smaller keys) << 16 | index: comparing keys is comparing the high halves.  The
// depth limit of the introsort (2 log2 n levels, then comb sort on the sub-range; sorted inputs reach it) is not reproduced in
// parallel: wave_qs_phase returns false and the caller runs the sequential form.  tests: bwahip_kat_introsort against the
// one-lane restatement of ksort.h on random lists with few and many ties.
#pragma once
#include "bwahip_internal.h"

namespace wv {

__device__ __forceinline__ void is_sync() { __threadfence_block(); __syncthreads(); }
__device__ __forceinline__ int is_wmax(int v) { for (int d = 32; d; d >>= 1) { const int o = __shfl_xor(v, d); v = v > o ? v : o; } return v; }
__device__ __forceinline__ unsigned long long is_le_mask(int l) { return l >= 63 ? ~0ull : ((2ull << l) - 1); }   // bits 0..l
__device__ __forceinline__ unsigned long long is_gt_shift(unsigned long long m, int l) { return l >= 63 ? 0ull : m >> (l + 1); }   // bits above l, shifted down

// median of three as ksort.h:193-196 picks it; returns the position of the pivot among s, t and the middle
__device__ __forceinline__ int is_pick(unsigned vi, unsigned vj, unsigned vk, int s, int t, int k)
{
	if ((vk >> 16) < (vi >> 16)) { if ((vk >> 16) < (vj >> 16)) k = t; }
	else k = (vj >> 16) < (vi >> 16) ? s : t;
	return k;
}

// One partition of v[s..t] (ksort.h:192-206) with t - s + 1 <= 64: every position has a lane.  tab: 128 words of LDS.  Returns i.
__device__ __forceinline__ int is_part_small(unsigned *v, int s, int t, unsigned *tab, int l)
{
	const int m = t - s + 1;
	const bool in = l < m;
	const int p = s + l;
	unsigned val = in ? v[p] : 0u;
	const int kmid = s + ((t - s) >> 1) + 1;
	const unsigned vi = __shfl(val, 0), vj = __shfl(val, m - 1), vk = __shfl(val, kmid - s);
	const int k = is_pick(vi, vj, vk, s, t, kmid);
	const unsigned rpv = k == s ? vi : k == t ? vj : vk;
	if (k != t) { if (p == k) val = vj; if (p == t) val = rpv; }     // the pivot goes to t
	const unsigned rr = rpv >> 16, r = val >> 16;
	const bool isL = in && l >= 1 && !(r < rr);
	const bool isR = in && l <= m - 2 && !(rr < r);
	const unsigned long long Lm = __ballot(isL), Rm = __ballot(isR);
	const int A = __popcll(Lm & is_le_mask(l)), B = __popcll(is_gt_shift(Rm, l));
	const int K = is_wmax(in ? (A < B ? A : B) : 0);
	const int rankL = __popcll(Lm & ((1ull << l) - 1)) + 1;            // 1-based from the left
	const int rankR = __popcll(is_gt_shift(Rm, l)) + 1;                // 1-based from the right
	const bool swL = isL && rankL <= K, swR = isR && rankR <= K;
	if (swL) tab[rankL - 1] = val;
	if (swR) tab[64 + rankR - 1] = val;
	is_sync();
	if (swL) val = tab[64 + rankL - 1];
	if (swR) val = tab[rankR - 1];
	int i;
	{
		const unsigned long long mL1 = __ballot(isL && rankL == K + 1), mRK = __ballot(isR && rankR == K);
		const int lK1 = mL1 ? __ffsll((long long)mL1) - 1 : 64;          // (exists: l_{K+1} <= t)
		const int rK = mRK ? __ffsll((long long)mRK) - 1 : 64;
		i = s + (K >= 1 ? (lK1 < rK ? lK1 : rK) : lK1);
	}
	{
		const unsigned v_i = __shfl(val, i - s), v_t = __shfl(val, m - 1);
		if (p == i) val = v_t;
		if (p == t) val = v_i;
	}
	if (in) v[p] = val;
	is_sync();
	return i;
}

// The same for longer ranges: flags chunk by chunk (64 positions each), masks and counts in scratch.  mL / mR: one word per chunk;
// cL[c] = scan-stops of the upward scan left of chunk c, cR[c] = stops of the downward scan right of chunk c; tabL / tabR: the positions to swap.
__device__ int is_part_big(unsigned *v, int s, int t, unsigned long long *mL, unsigned long long *mR, int *cL, int *cR, int *tabL, int *tabR, int l)
{
	const int m = t - s + 1, C = (m + 63) >> 6;
	const int kmid = s + ((t - s) >> 1) + 1;
	const unsigned vi = v[s], vj = v[t], vk = v[kmid];
	const int k = is_pick(vi, vj, vk, s, t, kmid);
	const unsigned rpv = k == s ? vi : k == t ? vj : vk;
	if (k != t) { if (l == 0) { v[k] = vj; v[t] = rpv; } }
	is_sync();
	const unsigned rr = rpv >> 16;
	for (int c = 0; c < C; ++c) {
		const int b = (c << 6) + l, p = s + b;
		const bool in = b < m;
		const unsigned r = in ? v[p] >> 16 : 0u;
		const unsigned long long Lm = __ballot(in && b >= 1 && !(r < rr)), Rm = __ballot(in && b <= m - 2 && !(rr < r));
		if (l == 0) { mL[c] = Lm; mR[c] = Rm; }
	}
	is_sync();
	// counts left / right of every chunk (serial over the chunks on every lane: C <= 1024, and only the first levels of a long list come here)
	if (l == 0) {
		int run = 0;
		for (int c = 0; c < C; ++c) { cL[c] = run; run += __popcll(mL[c]); }
		run = 0;
		for (int c = C - 1; c >= 0; --c) { cR[c] = run; run += __popcll(mR[c]); }
	}
	is_sync();
	int K = 0;
	for (int c = 0; c < C; ++c) {
		const int b = (c << 6) + l;
		const int A = cL[c] + __popcll(mL[c] & is_le_mask(l)), B = cR[c] + __popcll(is_gt_shift(mR[c], l));
		const int mn = b < m ? (A < B ? A : B) : 0;
		K = K > mn ? K : mn;
	}
	K = is_wmax(K);
	int lK1 = 0x7fffffff, rK = 0x7fffffff;
	for (int c = 0; c < C; ++c) {
		const int b = (c << 6) + l, p = s + b;
		const unsigned long long Lm = mL[c], Rm = mR[c];
		const bool isL = (Lm >> l) & 1, isR = (Rm >> l) & 1;
		const int rankL = cL[c] + __popcll(Lm & ((1ull << l) - 1)) + 1, rankR = cR[c] + __popcll(is_gt_shift(Rm, l)) + 1;
		if (isL && rankL <= K) tabL[rankL - 1] = p;
		if (isR && rankR <= K) tabR[rankR - 1] = p;
		const unsigned long long a1 = __ballot(isL && rankL == K + 1), a2 = __ballot(isR && rankR == K);
		if (a1) lK1 = s + (c << 6) + __ffsll((long long)a1) - 1;
		if (a2) rK = s + (c << 6) + __ffsll((long long)a2) - 1;
	}
	is_sync();
	for (int q = l; q < K; q += 64) {
		const int pl = tabL[q], pr = tabR[q];
		const unsigned a = v[pl], b = v[pr];
		v[pl] = b; v[pr] = a;
	}
	is_sync();
	const int i = K >= 1 ? (lK1 < rK ? lK1 : rK) : lK1;
	if (l == 0) { const unsigned a = v[i], b = v[t]; v[i] = b; v[t] = a; }
	is_sync();
	return i;
}

// ints of scratch wave_qs_phase needs for n elements (beside v itself)
__device__ __forceinline__ size_t is_scratch_ints(int n) { return 6 * ((size_t)n / 64 + 1) + 2 * ((size_t)n / 2 + 1); }

// The quicksort phase of ks_introsort on v[0..n) (packed words).  stk: 3 x 80 ints (LDS or global), tab: 128 words of LDS.
// false: a range reached the depth limit (the reference switches to comb sort there) -- v is then in an intermediate state.
__device__ bool wave_qs_phase(unsigned *v, int n, int *scratch, int *stk, unsigned *tab, int l)
{
	if (n < 2) return true;
	if (n == 2) {                                               // ksort.h:183-186
		const unsigned a = v[0], b = v[1];
		is_sync();
		if ((b >> 16) < (a >> 16) && l == 0) { v[0] = b; v[1] = a; }
		is_sync();
		return true;
	}
	// scratch (8-byte aligned): masks 2 x C1 words of 64 bits | counts 2 x C1 ints | tabL, tabR: n / 2 + 1 ints each
	const int C1 = n / 64 + 1;
	unsigned long long *mL = reinterpret_cast<unsigned long long*>(scratch), *mR = mL + C1;
	int *cL = reinterpret_cast<int*>(mR + C1), *cR = cL + C1;
	int *tabL = cR + C1, *tabR = tabL + n / 2 + 1;
	int d = 2;
	while ((1 << d) < n) ++d;
	d <<= 1;
	int s = 0, t = n - 1, top = 0;
	for (;;) {
		if (s < t) {
			if (--d == 0) return false;
			const int i = t - s + 1 <= 64 ? is_part_small(v, s, t, tab, l) : is_part_big(v, s, t, mL, mR, cL, cR, tabL, tabR, l);
			if (i - s > t - i) {
				if (i - s > 16) { if (l == 0) { stk[3 * top] = s; stk[3 * top + 1] = i - 1; stk[3 * top + 2] = d; } ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { if (l == 0) { stk[3 * top] = i + 1; stk[3 * top + 1] = t; stk[3 * top + 2] = d; } ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) return true;
			is_sync();
			--top; s = stk[3 * top]; t = stk[3 * top + 1]; d = stk[3 * top + 2];
		}
	}
}

// The final insertion sort as a stable placement: the element at position p of v goes to slot (its count of smaller keys) + (equal keys at
// positions before p) = the number of packed words (count << 16 | position) below its own.  tied[e] != 0: element e shares its key with
// another one (only those need the count; the others' slot is their count of smaller keys).  idx_out[slot] = element.  tile: 256 words of LDS.
__device__ void wave_final_place(const unsigned *v, int n, const uint8_t *tied, int *idx_out, unsigned *tile, int l)
{
	for (int base = 0; base < n; base += 64) {
		const int p = base + l;
		const unsigned val = p < n ? v[p] : 0u;
		const bool need = p < n && tied[val & 0xffffu];
		const unsigned mine = (val & 0xffff0000u) | (unsigned)p;
		int slot = (int)(val >> 16);
		if (__ballot(need)) {
			slot = 0;
			for (int qb = 0; qb < n; qb += 256) {
				is_sync();
				for (int q = qb + l; q < qb + 256 && q < n; q += 64) tile[q - qb] = (v[q] & 0xffff0000u) | (unsigned)q;
				is_sync();
				const int hi = n - qb < 256 ? n - qb : 256;
				for (int q = 0; q < hi; ++q) slot += tile[q] < mine ? 1 : 0;
			}
			if (!need) slot = (int)(val >> 16);
		}
		if (p < n) idx_out[slot] = (int)(val & 0xffffu);
	}
	is_sync();
}

} // namespace wv
