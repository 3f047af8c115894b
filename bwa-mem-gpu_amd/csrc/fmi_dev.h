// Device-side FM-index primitives for gfx950.
//
// Work shape: one Occ block (64 B = 4 x u64 counts + 128 bases, bwt.h:74-75; the bases as bit planes, see count_bases64) is fetched by a
// *quad* of 4 adjacent lanes, 16 B per lane (`global_load_dwordx4`), so every HBM request is one
// fully used, naturally aligned 64-byte line.  Lanes 0/1 of the quad hold the four base counts,
// lanes 2/3 hold 64 bases each and pop-count them; the partial results are combined with DPP
// quad permutes (no LDS traffic).  The reference computes the same numbers with a 256-entry byte
// table (bwt.c:42-51, 165-186); results are identical.
#pragma once
#include "bwahip_internal.h"

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP helpers (all VALU, no LDS): broadcast lane K of each quad / swap the two quads of an 8-lane group.
template <int K> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
	constexpr int ctrl = K | K << 2 | K << 4 | K << 6;      // quad_perm:[K,K,K,K]
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);
}
template <int K> __device__ __forceinline__ uint64_t quad_bcast64(uint64_t v)
{
	return (uint64_t)quad_bcast<K>((uint32_t)v) | (uint64_t)quad_bcast<K>((uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ uint32_t quad_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); } // [1,0,3,2]
__device__ __forceinline__ uint32_t quad_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); } // [2,3,0,1]
// row_half_mirror: lane i <-> 7-i inside each group of 8 lanes.  For values that are uniform inside a
// quad this hands each quad the value of the other quad of its 8-lane group.
__device__ __forceinline__ uint32_t half_mirror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); }
__device__ __forceinline__ uint64_t half_mirror64(uint64_t v)
{
	return (uint64_t)half_mirror((uint32_t)v) | (uint64_t)half_mirror((uint32_t)(v >> 32)) << 32;
}

// The 128 bases of a block IN HBM are kept as bit planes, not as bwa's 2-bit codes (k_bwt_planes re-lays them when the index is loaded; the
// files stay bwa's): the third 16 bytes of a block = {H0, H1, L0, L1}, the fourth = {H2, H3, L2, L3}, where Hg / Lg hold the high / low bit of
// the bases 32g .. 32g+31, first base in the top bit.  Counting the bases before a position is then a mask and three pop-counts per word pair
// (both bits, high, low) instead of separating the planes of 2-bit codes at every look-up: a third of k_smem's vector instructions were that.
//
// Count, for each base c, the symbols equal to c among the first `n` (0..64) of the 64 bases held in v = {H0, H1, L0, L1}.  Returns 4 byte counters.
__device__ __forceinline__ uint32_t count_bases64(uint4 v, int n)
{
	const int n0 = n < 32 ? n : 32, n1 = n - n0;
	const uint32_t m0 = (uint32_t)(0xffffffff00000000ull >> n0), m1 = (uint32_t)(0xffffffff00000000ull >> n1);   // the top n0 / n1 bits
	const uint32_t h0 = v.x & m0, h1 = v.y & m1, l0 = v.z & m0, l1 = v.w & m1;
	const uint32_t c3 = __popc(h0 & l0) + __popc(h1 & l1);
	const uint32_t ch = __popc(h0) + __popc(h1), cl = __popc(l0) + __popc(l1);
	const uint32_t c2 = ch - c3, c1 = cl - c3, c0 = (uint32_t)n - ch - cl + c3;
	return c0 | c1 << 8 | c2 << 16 | c3 << 24;
}
// the base at offset o (0..63) of the 64 bases held in v = {H0, H1, L0, L1}
__device__ __forceinline__ uint32_t base_at64(uint4 v, int o)
{
	const uint32_t hw = o < 32 ? v.x : v.y, lw = o < 32 ? v.z : v.w;
	const int bit = ~o & 31;
	return (hw >> bit & 1) << 1 | (lw >> bit & 1);
}

// occ4 for one position per quad (bwt.c:169 bwt_occ4).  `p` must be uniform inside the quad.
// All four lanes of the quad receive cnt[0..3].  `live` = false skips the load (result unspecified).
__device__ __forceinline__ void quad_occ4(const DevIndex &ix, uint64_t p, bool live, uint64_t cnt[4])
{
	const int r = lane_id() & 3;
	const bool none = (p == ~0ull);                          // bwt.c:173
	uint64_t pp = p - (p >= ix.primary);                     // '$' is not stored (bwt.c:177)
	uint4 v = make_uint4(0, 0, 0, 0);
	if (live && !none) v = ix.bwt[(pp >> 7) * 4 + r];
	int o = (int)(pp & 127) + 1;                             // bases 0..o-1 of the block are counted
	int n = r == 2 ? (o < 64 ? o : 64) : r == 3 ? (o > 64 ? o - 64 : 0) : 0;
	uint32_t packed = r >= 2 ? count_bases64(v, n) : 0u;
	packed += quad_xor1(packed);
	packed += quad_xor2(packed);                             // <=128 per byte: no carry between bytes
	uint64_t a = (uint64_t)v.y << 32 | v.x, b = (uint64_t)v.w << 32 | v.z;
	cnt[0] = quad_bcast64<0>(a) + (packed & 0xff);
	cnt[1] = quad_bcast64<0>(b) + (packed >> 8 & 0xff);
	cnt[2] = quad_bcast64<1>(a) + (packed >> 16 & 0xff);
	cnt[3] = quad_bcast64<1>(b) + (packed >> 24);
	if (none) cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
}

struct Bi { uint64_t x0, x1, x2; };                          // bi-interval without `info`

// Selects without dynamic indexing (dynamic indexing of kernel-argument arrays or local arrays would
// send them to scratch memory).
__device__ __forceinline__ uint64_t sel4(int c, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3)
{
	return c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3;
}
__device__ __forceinline__ uint64_t L2_at(const DevIndex &ix, int c)      // c in 0..4
{
	return c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : c == 3 ? ix.L2[3] : ix.L2[4];
}
__device__ __forceinline__ void set_intv(const DevIndex &ix, int b, Bi &ik)   // bwt_set_intv, bwt.h:82
{
	ik.x0 = L2_at(ix, b) + 1; ik.x2 = L2_at(ix, b + 1) - L2_at(ix, b); ik.x1 = L2_at(ix, 3 - b) + 1;
}

// bwt_extend (bwt.c:262) by an 8-lane group: quad 0 fetches the block of k, quad 1 the block of l.
// `ik`, `is_back`, `live` must be uniform inside the 8-lane group; every lane receives ok[0..3].
// Returns the number of distinct Occ blocks the reference would touch for this call (1 or 2; 0 if !live).
__device__ __forceinline__ int group8_extend(const DevIndex &ix, const Bi &ik, int is_back, bool live, Bi ok[4])
{
	const int quad = (lane_id() >> 2) & 1;
	uint64_t xa = is_back ? ik.x0 : ik.x1;                   // x[!is_back]
	uint64_t xb = is_back ? ik.x1 : ik.x0;                   // x[is_back]
	uint64_t k = xa - 1, l = k + ik.x2;
	uint64_t mine[4], other[4];
	quad_occ4(ix, quad ? l : k, live, mine);
#pragma unroll
	for (int i = 0; i < 4; ++i) other[i] = half_mirror64(mine[i]);
	uint64_t sz[4], lo[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		uint64_t tk = quad ? other[i] : mine[i], tl = quad ? mine[i] : other[i];
		lo[i] = ix.L2[i] + 1 + tk;
		sz[i] = tl - tk;
	}
	uint64_t b3 = xb + (xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary);
	uint64_t b2 = b3 + sz[3], b1 = b2 + sz[2], b0 = b1 + sz[1];
	uint64_t bb[4] = { b0, b1, b2, b3 };
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		ok[i].x0 = is_back ? lo[i] : bb[i];
		ok[i].x1 = is_back ? bb[i] : lo[i];
		ok[i].x2 = sz[i];
	}
	if (!live) return 0;
	uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	return (k == ~0ull || (kk >> 7) != (ll >> 7)) ? 2 : 1;   // bwt.c:194
}

// Same, but returns only ok[c] (c uniform inside the group, known before the call in every SMEM state).
__device__ __forceinline__ int group8_extend_c(const DevIndex &ix, const Bi &ik, int is_back, int c, bool live, Bi &o)
{
	const int quad = (lane_id() >> 2) & 1;
	uint64_t xa = is_back ? ik.x0 : ik.x1;
	uint64_t xb = is_back ? ik.x1 : ik.x0;
	uint64_t k = xa - 1, l = k + ik.x2;
	uint64_t mine[4];
	quad_occ4(ix, quad ? l : k, live, mine);
	uint64_t o0 = half_mirror64(mine[0]), o1 = half_mirror64(mine[1]), o2 = half_mirror64(mine[2]), o3 = half_mirror64(mine[3]);
	uint64_t tk0 = quad ? o0 : mine[0], tk1 = quad ? o1 : mine[1], tk2 = quad ? o2 : mine[2], tk3 = quad ? o3 : mine[3];
	uint64_t tl0 = quad ? mine[0] : o0, tl1 = quad ? mine[1] : o1, tl2 = quad ? mine[2] : o2, tl3 = quad ? mine[3] : o3;
	uint64_t s1 = tl1 - tk1, s2 = tl2 - tk2, s3 = tl3 - tk3;
	uint64_t lo = L2_at(ix, c) + 1 + sel4(c, tk0, tk1, tk2, tk3);
	uint64_t sz = sel4(c, tl0 - tk0, s1, s2, s3);
	uint64_t cum = (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
	uint64_t bb = xb + (xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) + cum;
	o.x0 = is_back ? lo : bb;
	o.x1 = is_back ? bb : lo;
	o.x2 = sz;
	if (!live) return 0;
	uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	return (k == ~0ull || (kk >> 7) != (ll >> 7)) ? 2 : 1;
}

// bwt_extend by ONE quad (16 reads per wavefront): the quad fetches the block of k and then the block of l
// (two independent 16-byte loads per lane, both in flight together).  Same result as group8_extend_c.
__device__ __forceinline__ int quad_extend_c(const DevIndex &ix, const Bi &ik, int is_back, int c, bool live, Bi &o)
{
	uint64_t xa = is_back ? ik.x0 : ik.x1;
	uint64_t xb = is_back ? ik.x1 : ik.x0;
	uint64_t k = xa - 1, l = k + ik.x2;
	uint64_t tk[4], tl[4];
	quad_occ4(ix, k, live, tk);
	quad_occ4(ix, l, live, tl);
	uint64_t s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
	uint64_t lo = L2_at(ix, c) + 1 + sel4(c, tk[0], tk[1], tk[2], tk[3]);
	uint64_t sz = sel4(c, tl[0] - tk[0], s1, s2, s3);
	uint64_t cum = (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
	uint64_t bb = xb + (xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) + cum;
	o.x0 = is_back ? lo : bb;
	o.x1 = is_back ? bb : lo;
	o.x2 = sz;
	if (!live) return 0;
	uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	return (k == ~0ull || (kk >> 7) != (ll >> 7)) ? 2 : 1;
}

// occ4 by ONE lane: the lane reads its own 64-byte block (4 x 16 B, every fetched byte used) and counts alone.
__device__ __forceinline__ void lane_occ4(const DevIndex &ix, uint64_t p, bool live, uint64_t cnt[4])
{
	const bool none = (p == ~0ull);
	uint64_t pp = p - (p >= ix.primary);
	uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
	if (live && !none) {
		const uint4 *b = ix.bwt + (pp >> 7) * 4;
		v0 = b[0]; v1 = b[1]; v2 = b[2]; v3 = b[3];
	}
	const int o = (int)(pp & 127) + 1;
	const uint32_t packed = count_bases64(v2, o < 64 ? o : 64) + count_bases64(v3, o > 64 ? o - 64 : 0);
	cnt[0] = ((uint64_t)v0.y << 32 | v0.x) + (packed & 0xff);
	cnt[1] = ((uint64_t)v0.w << 32 | v0.z) + (packed >> 8 & 0xff);
	cnt[2] = ((uint64_t)v1.y << 32 | v1.x) + (packed >> 16 & 0xff);
	cnt[3] = ((uint64_t)v1.w << 32 | v1.z) + (packed >> 24);
	if (none) cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
}
// occ4 of k and of l by one lane; when both fall into the same 64-byte block (two out of five extends) the block is fetched once:
// four load instructions fewer for those lanes (k_smem 28.9 -> 26.6 ms)
__device__ __forceinline__ void lane_occ4_pair(const DevIndex &ix, uint64_t pk, uint64_t pl, bool live, uint64_t ck[4], uint64_t cl[4])
{
	const bool nk = (pk == ~0ull);
	const uint64_t kk = pk - (pk >= ix.primary), ll = pl - (pl >= ix.primary);
	uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0;
	// the second 64 bases of a block are fetched only by the lanes whose position lies there
	const bool same = !nk && (kk >> 7) == (ll >> 7);
	const bool k_hi = (kk & 127) >= 64, l_hi = (ll & 127) >= 64;
	if (live && !nk) { const uint4 *b = ix.bwt + (kk >> 7) * 4; a0 = b[0]; a1 = b[1]; a2 = b[2]; if (k_hi || (same && l_hi)) a3 = b[3]; }
	uint4 b0 = a0, b1 = a1, b2 = a2, b3 = a3;
	if (live && !same && pl != ~0ull) { const uint4 *b = ix.bwt + (ll >> 7) * 4; b0 = b[0]; b1 = b[1]; b2 = b[2]; if (l_hi) b3 = b[3]; }   // (l = -1: the counts are zeroed below, nothing to fetch)
	{
		const int o = (int)(kk & 127) + 1;
		const uint32_t packed = count_bases64(a2, o < 64 ? o : 64) + count_bases64(a3, o > 64 ? o - 64 : 0);
		ck[0] = ((uint64_t)a0.y << 32 | a0.x) + (packed & 0xff); ck[1] = ((uint64_t)a0.w << 32 | a0.z) + (packed >> 8 & 0xff);
		ck[2] = ((uint64_t)a1.y << 32 | a1.x) + (packed >> 16 & 0xff); ck[3] = ((uint64_t)a1.w << 32 | a1.z) + (packed >> 24);
		if (nk) ck[0] = ck[1] = ck[2] = ck[3] = 0;
	}
	{
		const int o = (int)(ll & 127) + 1;
		const uint32_t packed = count_bases64(b2, o < 64 ? o : 64) + count_bases64(b3, o > 64 ? o - 64 : 0);
		cl[0] = ((uint64_t)b0.y << 32 | b0.x) + (packed & 0xff); cl[1] = ((uint64_t)b0.w << 32 | b0.z) + (packed >> 8 & 0xff);
		cl[2] = ((uint64_t)b1.y << 32 | b1.x) + (packed >> 16 & 0xff); cl[3] = ((uint64_t)b1.w << 32 | b1.z) + (packed >> 24);
		if (pl == ~0ull) cl[0] = cl[1] = cl[2] = cl[3] = 0;
	}
}

// bwt_extend by ONE lane (64 reads per wavefront): 2 x 64 B per lane in flight, no cross-lane traffic at all.
__device__ __forceinline__ int lane_extend_c(const DevIndex &ix, const Bi &ik, int is_back, int c, bool live, Bi &o)
{
	uint64_t xa = is_back ? ik.x0 : ik.x1;
	uint64_t xb = is_back ? ik.x1 : ik.x0;
	uint64_t k = xa - 1, l = k + ik.x2;
	uint64_t tk[4], tl[4];
	lane_occ4_pair(ix, k, l, live, tk, tl);
	uint64_t s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
	uint64_t lo = L2_at(ix, c) + 1 + sel4(c, tk[0], tk[1], tk[2], tk[3]);
	uint64_t sz = sel4(c, tl[0] - tk[0], s1, s2, s3);
	uint64_t cum = (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
	uint64_t bb = xb + (xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) + cum;
	o.x0 = is_back ? lo : bb;
	o.x1 = is_back ? bb : lo;
	o.x2 = sz;
	if (!live) return 0;
	uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	return (k == ~0ull || (kk >> 7) != (ll >> 7)) ? 2 : 1;
}

// bwt_extend by a PAIR of lanes (32 reads per wavefront): lane 0 of the pair reads and counts the whole block of k,
// lane 1 the block of l; one DPP exchange hands each the other's four counts.
__device__ __forceinline__ uint64_t pair_swap64(uint64_t v) { return (uint64_t)quad_xor1((uint32_t)v) | (uint64_t)quad_xor1((uint32_t)(v >> 32)) << 32; }
__device__ __forceinline__ int pair_extend_c(const DevIndex &ix, const Bi &ik, int is_back, int c, bool live, Bi &o)
{
	const int h = lane_id() & 1;
	uint64_t xa = is_back ? ik.x0 : ik.x1;
	uint64_t xb = is_back ? ik.x1 : ik.x0;
	uint64_t k = xa - 1, l = k + ik.x2;
	uint64_t mine[4];
	lane_occ4(ix, h ? l : k, live, mine);
	uint64_t o0 = pair_swap64(mine[0]), o1 = pair_swap64(mine[1]), o2 = pair_swap64(mine[2]), o3 = pair_swap64(mine[3]);
	uint64_t tk0 = h ? o0 : mine[0], tk1 = h ? o1 : mine[1], tk2 = h ? o2 : mine[2], tk3 = h ? o3 : mine[3];
	uint64_t tl0 = h ? mine[0] : o0, tl1 = h ? mine[1] : o1, tl2 = h ? mine[2] : o2, tl3 = h ? mine[3] : o3;
	uint64_t s1 = tl1 - tk1, s2 = tl2 - tk2, s3 = tl3 - tk3;
	uint64_t lo = L2_at(ix, c) + 1 + sel4(c, tk0, tk1, tk2, tk3);
	uint64_t sz = sel4(c, tl0 - tk0, s1, s2, s3);
	uint64_t cum = (c < 3 ? s3 : 0) + (c < 2 ? s2 : 0) + (c < 1 ? s1 : 0);
	uint64_t bb = xb + (xa <= ix.primary && xa + ik.x2 - 1 >= ix.primary) + cum;
	o.x0 = is_back ? lo : bb;
	o.x1 = is_back ? bb : lo;
	o.x2 = sz;
	if (!live) return 0;
	uint64_t kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	return (k == ~0ull || (kk >> 7) != (ll >> 7)) ? 2 : 1;
}

// One LF step (bwt.c:53 bwt_invPsi) by ONE lane: the lane reads the whole 64-byte block of k.
// lane_lf in two halves -- the block fetch and the count -- so that a lane walking several rows at once (k_seed_walk) has all its fetches
// in flight before it waits for the first
struct LfBlock { uint4 v0, v1, v2, v3; };
__device__ __forceinline__ void lane_lf_fetch(const DevIndex &ix, uint64_t k, LfBlock &f)
{
	const uint64_t pp = k - (k >= ix.primary);
	const uint4 *b = ix.bwt + (pp >> 7) * 4;
	f.v3 = make_uint4(0, 0, 0, 0);
	f.v0 = b[0]; f.v1 = b[1]; f.v2 = b[2];
	if ((pp & 127) >= 64) f.v3 = b[3];                       // the second 64 bases only when the position lies there (k_seed_walk 8.3 -> 8.0 ms)
}
__device__ __forceinline__ uint64_t lane_lf_count(const DevIndex &ix, uint64_t k, const LfBlock &f)
{
	const uint64_t pp = k - (k >= ix.primary);
	const int o = (int)(pp & 127);
	const uint32_t c = base_at64(o < 64 ? f.v2 : f.v3, o & 63);
	const uint32_t packed = count_bases64(f.v2, o + 1 < 64 ? o + 1 : 64) + count_bases64(f.v3, o + 1 > 64 ? o + 1 - 64 : 0);
	const uint64_t c0 = (uint64_t)f.v0.y << 32 | f.v0.x, c1 = (uint64_t)f.v0.w << 32 | f.v0.z, c2 = (uint64_t)f.v1.y << 32 | f.v1.x, c3 = (uint64_t)f.v1.w << 32 | f.v1.z;
	const uint64_t x = L2_at(ix, (int)c) + sel4((int)c, c0, c1, c2, c3) + (packed >> (c << 3) & 0xff);
	return k == ix.primary ? 0 : x;
}
__device__ __forceinline__ uint64_t lane_lf(const DevIndex &ix, uint64_t k)
{
	LfBlock f;
	lane_lf_fetch(ix, k, f);
	return lane_lf_count(ix, k, f);
}

// One LF step (bwt.c:53 bwt_invPsi) by a quad; k uniform inside the quad; all lanes get the result.
__device__ __forceinline__ uint64_t quad_lf(const DevIndex &ix, uint64_t k)
{
	const int r = lane_id() & 3;
	uint64_t pp = k - (k >= ix.primary);                     // for k != primary identical to k - (k > primary)
	uint4 v = ix.bwt[(pp >> 7) * 4 + r];
	int o = (int)(pp & 127);                                 // symbol at offset o; occ counts bases 0..o
	// the symbol: lane 2 has the planes of bases 0-63, lane 3 of 64-127
	uint32_t sym = base_at64(v, o & 63);
	uint32_t s2 = quad_bcast<2>(sym), s3 = quad_bcast<3>(sym);
	uint32_t c = o < 64 ? s2 : s3;
	int n = r == 2 ? (o + 1 < 64 ? o + 1 : 64) : r == 3 ? (o + 1 > 64 ? o + 1 - 64 : 0) : 0;
	uint32_t packed = r >= 2 ? count_bases64(v, n) : 0u;
	packed += quad_xor1(packed);
	packed += quad_xor2(packed);
	uint64_t a = (uint64_t)v.y << 32 | v.x, b = (uint64_t)v.w << 32 | v.z;
	uint64_t base = c == 0 ? quad_bcast64<0>(a) : c == 1 ? quad_bcast64<0>(b) : c == 2 ? quad_bcast64<1>(a) : quad_bcast64<1>(b);
	uint64_t x = ix.L2[c] + base + (packed >> (c << 3) & 0xff);
	return k == ix.primary ? 0 : x;
}
