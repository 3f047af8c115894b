// Sorting a read's region list the way the reference does (shared by k_extend.hip: mem_sort_dedup_patch, and k_pair.hip:
// mem_sort_dedup_patch as mem_matesw calls it): ks_introsort on mem_ars2 / mem_ars (bwamem.c:398-402, ksort.h:176-227).
#pragma once
#include "bwahip_internal.h"
#include "isort_dev.h"

namespace wv {

// ---- comparators of bwamem.c:398-402 on an index array into the read's DevReg list; exact introsort ----
// The sort keys are copied out of the 80-byte records into a dense 16-byte array first (by all lanes): the sorts run on
// one lane, and what they wait for is the latency of the key fetches.
struct RegKey { int64_t k64; int score, qb; };               // mode 0: k64 = re; mode 1: k64 = rb
struct RegSort { const RegKey *key; int mode; };              // mode 0: by re (mem_ars2); 1: score desc, rb, qb (mem_ars)
__device__ __forceinline__ bool reg_lt(const RegSort s, int x, int y)
{
	const RegKey p = s.key[x], q = s.key[y];
	if (s.mode == 0) return p.k64 < q.k64;
	return p.score > q.score || (p.score == q.score && (p.k64 < q.k64 || (p.k64 == q.k64 && p.qb < q.qb)));
}
// Sort by ranks with the whole wavefront when no two keys are equal (then every correct sort, the reference's unstable
// introsort included, produces the same order); returns false, leaving idx untouched, as soon as a tie exists -- the
// caller then runs the exact introsort on one lane.  Only worth it for long lists.
__device__ __forceinline__ bool wave_rank_sort(const RegSort c, int n, int *idx, int l)
{
	for (int base = 0; base < n; base += 64) {
		const int t = base + l;
		int rank = 0;
		bool tie = false;
		if (t < n) {
			const RegKey kt = c.key[t];
			for (int u = 0; u < n; ++u) {
				const RegKey ku = c.key[u];
				bool lt_ut, eq;
				if (c.mode == 0) { lt_ut = ku.k64 < kt.k64; eq = ku.k64 == kt.k64; }
				else {
					eq = ku.score == kt.score && ku.k64 == kt.k64 && ku.qb == kt.qb;
					lt_ut = ku.score > kt.score || (ku.score == kt.score && (ku.k64 < kt.k64 || (ku.k64 == kt.k64 && ku.qb < kt.qb)));
				}
				rank += lt_ut ? 1 : 0;
				tie |= eq && u != t;
			}
		}
		if (__ballot(tie)) return false;                     // idx[0..n) has not been touched
		if (t < n) idx[n + rank] = t;                        // second half of idx is free (2 ints per seed slot)
	}
	__threadfence_block(); __syncthreads();
	for (int i = l; i < n; i += 64) idx[i] = idx[n + i];
	return true;
}

// ---- helpers of the wavefront sort ----
__device__ __forceinline__ bool rk_lt(int mode, const RegKey &a, const RegKey &b)      // a sorts before b
{
	if (mode == 0) return a.k64 < b.k64;
	return a.score > b.score || (a.score == b.score && (a.k64 < b.k64 || (a.k64 == b.k64 && a.qb < b.qb)));
}
__device__ __forceinline__ bool rk_eq(int mode, const RegKey &a, const RegKey &b)
{
	return mode == 0 ? a.k64 == b.k64 : (a.score == b.score && a.k64 == b.k64 && a.qb == b.qb);
}
__device__ __forceinline__ RegKey rk_shfl(const RegKey &k, int src)
{
	RegKey o;
	o.k64 = (int64_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)k.k64, src) | (uint64_t)(uint32_t)__shfl((int)((uint64_t)k.k64 >> 32), src) << 32);
	o.score = __shfl(k.score, src); o.qb = __shfl(k.qb, src);
	return o;
}

// Count of smaller keys (<< 16 | index -> v[]) and "shares its key" (-> tied[]) of every element of a long list, by buckets: 64 sampled
// keys, sorted, cut the key space into 65 ranges; an element's count is the size of the ranges below its own plus its rank inside its
// range, and equal keys always share a range.  n x 64 comparisons to find the ranges and about n x n / 64 / 64 inside them, instead of
// n x n / 64 (a list of 3 000 regions -- one end of a pair inside a repeat family, re-sorted after every rescued hit -- took 3.5 ms per
// sort in the all-against-all count).  perm: n ints of scratch; lds: 256 words.  Returns a non-zero mask when any two keys are equal.
__device__ unsigned long long rank_pass_bucketed(const RegSort c, int n, unsigned *v, uint8_t *tied, int *perm, unsigned *lds, int l)
{
	const int mode = c.mode;
	RegKey *tile = reinterpret_cast<RegKey*>(lds);
	// the splitters: lane s samples one key; sorted across the lanes by (key, lane), lane j then holds the j-th smallest
	RegKey spl;
	{
		const RegKey ks = c.key[(long long)l * n / 64];
		int rs = 0;
		for (int u = 0; u < 64; ++u) { const RegKey ku = rk_shfl(ks, u); rs += (rk_lt(mode, ku, ks) || (rk_eq(mode, ku, ks) && u < l)) ? 1 : 0; }
		is_sync();
		tile[rs] = ks;
		is_sync();
		spl = tile[l];
		is_sync();
	}
	int *hist = reinterpret_cast<int*>(lds), *base = hist + 66, *fill = base + 66;      // 65 ranges: 0 .. 64
	for (int i = l; i < 3 * 66; i += 64) hist[i] = 0;
	is_sync();
	for (int tb = 0; tb < n; tb += 64) {                         // the range of every element = the number of splitters below its key
		const int t = tb + l;
		RegKey kt; kt.k64 = 0; kt.score = 0; kt.qb = 0;
		if (t < n) kt = c.key[t];
		int b = 0;
		for (int sp = 0; sp < 64; ++sp) { const RegKey ks = rk_shfl(spl, sp); b += rk_lt(mode, ks, kt) ? 1 : 0; }
		if (t < n) { v[t] = (unsigned)b; atomicAdd(&hist[b], 1); }
	}
	is_sync();
	{                                                            // sizes -> starts (65 values on 64 lanes: the last one by hand)
		const int h = hist[l];
		int inc = h;
		for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (l >= d) inc += o; }
		base[l] = inc - h;
		if (l == 63) base[64] = inc;
	}
	is_sync();
	for (int t = l; t < n; t += 64) { const int b = (int)v[t]; perm[base[b] + atomicAdd(&fill[b], 1)] = t; }
	is_sync();
	unsigned long long any_tie = 0;
	for (int b = 0; b <= 64; ++b) {                              // inside a range: its members one per lane, 64 at a time, the others by shuffle
		const int m = hist[b], start = base[b];
		for (int jb = 0; jb < m; jb += 64) {
			const int j = jb + l;
			int t = -1;
			RegKey kt; kt.k64 = 0; kt.score = 0; kt.qb = 0;
			if (j < m) { t = perm[start + j]; kt = c.key[t]; }
			int cnt = 0; bool tie = false;
			for (int ub = 0; ub < m; ub += 64) {
				RegKey ko; ko.k64 = 0; ko.score = 0; ko.qb = 0;
				if (ub == jb) ko = kt;
				else if (ub + l < m) ko = c.key[perm[start + ub + l]];
				const int hi = m - ub < 64 ? m - ub : 64;
				for (int u = 0; u < hi; ++u) {
					const RegKey ku = rk_shfl(ko, u);
					cnt += rk_lt(mode, ku, kt) ? 1 : 0;
					tie |= rk_eq(mode, ku, kt) && ub + u != j;
				}
			}
			if (j < m) { v[t] = (unsigned)(start + cnt) << 16 | (unsigned)t; tied[t] = tie ? 1 : 0; }
			any_tie |= __ballot(j < m && tie);
		}
	}
	is_sync();
	return any_tie;
}

// ks_introsort(mem_ars2 / mem_ars) on idx[0..n) (identity on entry) by the whole wavefront, exact for any input: keys without ties are
// placed by rank; with ties the quicksort phase of the reference's introsort is followed in parallel (isort_dev.h).
//   work: 8-byte aligned global scratch of ws_work_ints(n) ints; stk: 240 ints (LDS); lds: 256 words of LDS.
// false (idx untouched): the introsort's depth limit was reached or n > 65535 -- the caller runs rs_introsort on one lane.
__device__ __forceinline__ size_t ws_work_ints(int n) { return is_scratch_ints(n) + (size_t)n + (size_t)n / 4 + 4; }
//   v_lds (optional): v_cap words of LDS for the packed elements -- the partition steps then run on LDS instead of global memory
__device__ bool wave_sort_exact(const RegSort c, int n, int *idx, int *work, int *stk, unsigned *lds, int l, unsigned *v_lds = nullptr, int v_cap = 0)
{
	if (n > 65535) return false;
	if (n <= 64) {
		// short lists (nearly all of them: two or three regions): one key per lane, the others come by shuffle; without ties the count of
		// smaller keys is the slot and nothing but idx is touched
		RegKey kt; kt.k64 = 0; kt.score = 0; kt.qb = 0;
		if (l < n) kt = c.key[l];
		int cnt = 0; bool tie = false;
		for (int u = 0; u < n; ++u) {
			RegKey ku;
			ku.k64 = (int64_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)kt.k64, u) | (uint64_t)(uint32_t)__shfl((int)((uint64_t)kt.k64 >> 32), u) << 32);
			ku.score = __shfl(kt.score, u); ku.qb = __shfl(kt.qb, u);
			bool lt_ut, eq;
			if (c.mode == 0) { lt_ut = ku.k64 < kt.k64; eq = ku.k64 == kt.k64; }
			else {
				eq = ku.score == kt.score && ku.k64 == kt.k64 && ku.qb == kt.qb;
				lt_ut = ku.score > kt.score || (ku.score == kt.score && (ku.k64 < kt.k64 || (ku.k64 == kt.k64 && ku.qb < kt.qb)));
			}
			cnt += lt_ut ? 1 : 0;
			tie |= eq && u != l;
		}
		if (!__ballot(l < n && tie)) {
			if (l < n) idx[cnt] = l;
			is_sync();
			return true;
		}
	}
	int *qs_scratch = work;
	unsigned *v = n <= v_cap ? v_lds : reinterpret_cast<unsigned*>(work + is_scratch_ints(n));
	uint8_t *tied = reinterpret_cast<uint8_t*>(work + is_scratch_ints(n) + n);
	RegKey *tile = reinterpret_cast<RegKey*>(lds);               // 64 keys of 16 bytes
	unsigned long long any_tie = 0;
	if (n >= 384) any_tie = rank_pass_bucketed(c, n, v, tied, qs_scratch + 6 * (n / 64 + 1), lds, l);   // (perm: where the partition tables will be)
	else
	// count of smaller keys (and "has an equal one") of every element: four elements per lane and pass, the keys they are compared with
	// coming by in tiles of 64 through LDS -- n^2 / 64 comparisons per lane, n / 256 tile loads per 256 elements
	for (int base = 0; base < n; base += 256) {
		RegKey kt[4];
		int cnt[4] = { 0, 0, 0, 0 }; bool tie[4] = { false, false, false, false };
#pragma unroll
		for (int e = 0; e < 4; ++e) { const int t = base + 64 * e + l; kt[e].k64 = 0; kt[e].score = 0; kt[e].qb = 0; if (t < n) kt[e] = c.key[t]; }
		for (int ub = 0; ub < n; ub += 64) {
			is_sync();
			if (ub + l < n) tile[l] = c.key[ub + l];
			is_sync();
			const int hi = n - ub < 64 ? n - ub : 64;
			for (int u = 0; u < hi; ++u) {
				const RegKey ku = tile[u];
#pragma unroll
				for (int e = 0; e < 4; ++e) {
					bool lt_ut, eq;
					if (c.mode == 0) { lt_ut = ku.k64 < kt[e].k64; eq = ku.k64 == kt[e].k64; }
					else {
						eq = ku.score == kt[e].score && ku.k64 == kt[e].k64 && ku.qb == kt[e].qb;
						lt_ut = ku.score > kt[e].score || (ku.score == kt[e].score && (ku.k64 < kt[e].k64 || (ku.k64 == kt[e].k64 && ku.qb < kt[e].qb)));
					}
					cnt[e] += lt_ut ? 1 : 0;
					tie[e] |= eq && ub + u != base + 64 * e + l;
				}
			}
		}
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			const int t = base + 64 * e + l;
			if (t < n) { v[t] = (unsigned)cnt[e] << 16 | (unsigned)t; tied[t] = tie[e] ? 1 : 0; }
			any_tie |= __ballot(t < n && tie[e]);
		}
	}
	is_sync();
	if (!any_tie) {
		for (int t = l; t < n; t += 64) idx[v[t] >> 16] = t;
		is_sync();
		return true;
	}
	if (!wave_qs_phase(v, n, qs_scratch, stk, lds, l)) return false;
	wave_final_place(v, n, tied, idx, lds, l);
	return true;
}

__device__ __forceinline__ void rs_insertion(const RegSort c, int *s, int *t)
{
	for (int *i = s + 1; i < t; ++i)
		for (int *j = i; j > s && reg_lt(c, *j, *(j - 1)); --j) { int tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
__device__ __forceinline__ void rs_comb(const RegSort c, int n, int *a)
{
	const double shrink = 1.2473309501039786540366528676643;
	int swapped, gap = n;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = 0;
		for (int *i = a; i < a + n - gap; ++i) {
			int *j = i + gap;
			if (reg_lt(c, *j, *i)) { int tmp = *i; *i = *j; *j = tmp; swapped = 1; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) rs_insertion(c, a, a + n);
}
// `budget` bounds the work so that a logic error can never hang the GPU: on exhaustion the sort stops and
// the kernel reports BWAHIP_EINTERNAL (never expected; n log n comparisons suffice)
__device__ __forceinline__ void rs_introsort(const RegSort c, int n, int *a, int *stk, int *bad)   // ksort.h:176-227
{
	int d, top = 0, *s, *t, *i, *j, *k, pivot, tmp;
	long budget = 64L * n * 32 + 1024;
	if (n < 1) return;
	if (n == 2) { if (reg_lt(c, a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
	for (d = 2; 1 << d < n; ++d);
	s = a; t = a + (n - 1); d <<= 1;
	for (;;) {
		if (--budget < 0) { *bad = 1; return; }
		if (s < t) {
			if (--d == 0) { rs_comb(c, (int)(t - s) + 1, s); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (reg_lt(c, *k, *i)) { if (reg_lt(c, *k, *j)) k = j; }
			else k = reg_lt(c, *j, *i) ? i : j;
			pivot = *k;
			if (k != t) { tmp = *k; *k = *t; *t = tmp; }
			for (;;) {
				do ++i; while (i < t && reg_lt(c, *i, pivot));       // i stops at the pivot (at t) at the latest
				do --j; while (i <= j && reg_lt(c, pivot, *j));
				if (j <= i) break;
				if (--budget < 0) { *bad = 2; return; }
				tmp = *i; *i = *j; *j = tmp;
			}
			tmp = *i; *i = *t; *t = tmp;
			if (i - s > t - i) {
				if (i - s > 16) { stk[3*top] = (int)(s - a); stk[3*top+1] = (int)(i - 1 - a); stk[3*top+2] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stk[3*top] = (int)(i + 1 - a); stk[3*top+1] = (int)(t - a); stk[3*top+2] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { rs_insertion(c, a, a + n); return; }
			--top; s = a + stk[3*top]; t = a + stk[3*top+1]; d = stk[3*top+2];
		}
	}
}



} // namespace wv
