// Wavefront-level helpers shared by the DP kernels (k_extend.hip, k_final.hip): DPP scans / reductions, reference access.
#pragma once
#include "bwahip_internal.h"

namespace wv {

constexpr int NEG = -0x40000000;                             // MINUS_INF of ksw.c:489
constexpr int LOW = -0x60000000;                             // below every real DP value, and LOW - 1000*e stays above INT_MIN

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
// ---- wavefront reductions / scan on DPP (VALU only).  The tail of this kernel is one wavefront walking the
// rows of one read, so the dependent latency of a row matters: a 6-step ds_bpermute reduction costs ~700
// cycles, the same reduction on DPP row shifts / row broadcasts ~50.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_or(int v, int fill)
{
	return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);   // lanes without a source keep `fill`
}
// inclusive prefix max over the 64 lanes (lane i gets max of lanes 0..i).  A lane without a DPP source keeps `old`,
// and old = v makes that the identity of max, so every step is a single v_max_i32_dpp.
// One v_max_i32_dpp per step, in place: a lane without a DPP source is not written and so keeps its own value, the
// identity of max.  Written as asm because the builtin form compiles to v_mov + v_mov_dpp + v_max (3 x the issue slots);
// the s_nop 1 are the two wait states a DPP read of a just-written VGPR needs (the compiler cannot see into the asm).
__device__ __forceinline__ int wscan_incl_max(int v, int /*ident*/)
{
	asm volatile(
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
		"s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
		"s_nop 1"
		: "+v"(v));
	return v;
}
__device__ __forceinline__ int wmax(int v) { return __builtin_amdgcn_readlane(wscan_incl_max(v, (int)0x80000000), 63); }
__device__ __forceinline__ int wmin(int v) { return -wmax(-v); }          // callers never pass INT_MIN
__device__ __forceinline__ int wsum(int v) { for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d); return v; }
__device__ __forceinline__ int64_t wmax64(int64_t v) { for (int d = 32; d; d >>= 1) { int64_t o = __shfl_xor(v, d); v = v > o ? v : o; } return v; }
__device__ __forceinline__ int64_t wmin64(int64_t v) { for (int d = 32; d; d >>= 1) { int64_t o = __shfl_xor(v, d); v = v < o ? v : o; } return v; }
// exclusive prefix max over lanes (lane 0 gets `ident`)
__device__ __forceinline__ int wscan_excl_max(int v, int ident)
{
	const int inc = wscan_incl_max(v, ident);
	const int p = __builtin_amdgcn_update_dpp(ident, inc, 0x138, 0xf, 0xf, false);   // wave_shr:1
	return p;
}

// (int)((double)x / e + 1.) for integers x and e > 0, without leaving integer arithmetic: x / e is an integer plus r / e with
// 0 <= r < e, so the double quotient never sits within rounding distance of an integer it is not equal to, and the
// truncation toward zero is floor for x + e >= 0 and ceil below (ksw.c:402-407, bwamem.c:630-631).
__device__ __forceinline__ int div_plus1_trunc(int x, int e)
{
	const int y = x + e;
	if (e == 1) return y;
	return y >= 0 ? y / e : -((-y) / e);
}

struct Sw { const int8_t *mat; int o_del, e_del, o_ins, e_ins, mx; };   // mx = largest entry of mat (ksw.c:399)

__device__ __forceinline__ int dev_pos2rid(const DevIndex &ix, int64_t pos_f)   // bntseq.c:354
{
	if (pos_f >= ix.l_pac) return -1;
	int left = 0, mid = 0, right = ix.n_seqs;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= ix.anns[mid].offset) {
			if (mid == ix.n_seqs - 1) break;
			if (pos_f < ix.anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}
__device__ __forceinline__ int pac_at(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }
// base at coordinate p of the forward+reverse-complement reference (bns_get_seq, bntseq.c:403-424)
__device__ __forceinline__ int ref_base(const DevIndex &ix, int64_t p)
{
	return p < ix.l_pac ? pac_at(ix.pac, p) : 3 - pac_at(ix.pac, (ix.l_pac << 1) - 1 - p);
}

} // namespace wv
