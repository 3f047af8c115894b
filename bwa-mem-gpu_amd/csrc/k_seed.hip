// K2 -- seed positions: suffix-array look-up for every occurrence chaining will use
// (the `bwt_sa` call of mem_chain, bwamem.c:290, with the max_occ striding of bwamem.c:285-286),
// plus the contig id test of bwamem.c:293 (bns_intv2rid, bntseq.c:370).
//
// All look-ups of a batch are independent, so they are flattened: an exclusive scan of the per-read
// look-up counts (written by K1) gives every seed a fixed slot, in exactly the order the reference
// visits them (interval order, then k).  Three small kernels: k_seed_rows lays the BWT rows out in the slots,
// k_seed_walk resolves them (one lane per look-up), k_seed_rid adds the contig id.
// Also holds the known-answer kernels the parity tests drive (bwahip_kat_*).
#include "fmi_dev.h"

namespace {

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
__device__ __forceinline__ int64_t dev_depos(const DevIndex &ix, int64_t pos) { return pos >= ix.l_pac ? (ix.l_pac << 1) - 1 - pos : pos; }
__device__ __forceinline__ int dev_intv2rid(const DevIndex &ix, int64_t rb, int64_t re)   // bntseq.c:370
{
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	int rid_b = dev_pos2rid(ix, dev_depos(ix, rb));
	int rid_e = rb < re ? dev_pos2rid(ix, dev_depos(ix, re - 1)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

// bwt_sa (bwt.c:86) by a quad; `live` quads walk, the others only take part in the DPP exchanges.
__device__ __forceinline__ uint64_t quad_sa(const DevIndex &ix, uint64_t k, bool live, unsigned long long &n_lf)
{
	const uint64_t mask = (uint64_t)ix.sa_intv - 1;
	uint64_t steps = 0;
	for (;;) {
		bool go = live && (k & mask) != 0;
		if (__ballot(go) == 0) break;
		uint64_t safe = go && k != ix.primary ? k : 0;       // idle quads gather block 0 lanes (harmless, no fault)
		uint64_t nk = quad_lf(ix, safe);
		if (go) { k = (k == ix.primary) ? 0 : nk; ++steps; ++n_lf; }
	}
	return live ? steps + ix.sa[k >> ix.sa_shift] : 0;
}

// K2a -- one thread per read lays out its look-ups: slot seed_base[r] + running index holds the BWT row to resolve
// (kept in the rbeg field until K2b overwrites it) and the seed's query interval, in the order the reference visits
// them (interval order, then k; bwamem.c:283-290).
__global__ __launch_bounds__(256) void k_seed_rows(SeedLaunch a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.n_reads) return;
	const long long base = a.seed_base[r], end = a.seed_base[r + 1];
	if (end == base) return;                                 // also covers reads whose interval list overflowed
	const int n = a.intv_n[r];
	const DevIntv *iv = a.intv + (size_t)r * a.cap;
	long long sid = base;
	for (int t = 0; t < n; ++t) {
		const uint64_t x0 = iv[t].x0, x2 = iv[t].x2, info = iv[t].info;
		const uint64_t step = x2 > (uint64_t)a.opt.max_occ ? x2 / a.opt.max_occ : 1;   // bwamem.c:285
		DevSeed sd;
		sd.qbeg = (int)(info >> 32); sd.len = (int)(uint32_t)info - sd.qbeg; sd.score = sd.len; sd.rid = 0;
		int count = 0;
		for (uint64_t kk = 0; kk < x2 && count < a.opt.max_occ; kk += step, ++count) {   // bwamem.c:286
			sd.rbeg = (int64_t)(x0 + kk);
			a.seeds[sid++] = sd;
		}
	}
}

// K2b -- bwt_sa (bwt.c:86): one lane per look-up, every lane walking on its own: up to sa_intv-1 dependent LF steps
// (bwt.c:53), each one whole 64-byte Occ block read by the lane itself, then one 8-byte SA read.  A lane that
// finishes takes its next slot at once (the row was prefetched), so the wavefront never waits for its slowest walk.
// W walks per lane.  A walk is a chain of dependent 64-byte gathers, so a lane with one walk has one request in flight; W = 2 issues the
// second walk's request while the first is on its way.  Measured (3.1 Gbp index, 1 M-read batches, same box, twice): 8.42 ms with one walk,
// 8.80 ms with two -- 2 048 lanes per CU already keep about two thousand requests in flight, the kernel sits at the request rate of the
// memory system (scripts/gather_bw.hip), and the second walk only costs registers and issue slots.  W = 1 is what runs; BWAHIP_SEED_WALKS=2 selects the other.
template <int W>
__global__ __launch_bounds__(256) void k_seed_walk(SeedLaunch a, long long total)
{
	const DevIndex &ix = a.ix;
	const long long n_lanes = (long long)gridDim.x * blockDim.x;
	long long next = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t mask = (uint64_t)ix.sa_intv - 1;
	uint64_t k_pref = next < total ? (uint64_t)a.seeds[next].rbeg : 0;
	long long sid[W]; uint64_t k[W]; unsigned steps[W]; bool have[W];
#pragma unroll
	for (int w = 0; w < W; ++w) { sid[w] = 0; k[w] = 0; steps[w] = 0; have[w] = false; }
	unsigned n_lf = 0, n_sa = 0;
	for (;;) {
		bool any = false;
#pragma unroll
		for (int w = 0; w < W; ++w) {
			if (!have[w] && next < total) {
				sid[w] = next; k[w] = k_pref; steps[w] = 0; have[w] = true;
				next += n_lanes;
				if (next < total) k_pref = (uint64_t)a.seeds[next].rbeg;
			}
			any |= have[w];
		}
		if (__ballot(any) == 0) break;
		uint64_t sa_v[W]; bool fin[W], walk[W]; LfBlock blk[W];
#pragma unroll
		for (int w = 0; w < W; ++w) {                            // all gathers of this round are issued before any is used
			fin[w] = have[w] && (k[w] & mask) == 0;
			walk[w] = have[w] && !fin[w];
			sa_v[w] = 0;
			blk[w].v0 = blk[w].v1 = blk[w].v2 = blk[w].v3 = make_uint4(0, 0, 0, 0);
			if (fin[w]) sa_v[w] = ix.sa[k[w] >> ix.sa_shift];
			if (walk[w]) lane_lf_fetch(ix, k[w], blk[w]);
		}
#pragma unroll
		for (int w = 0; w < W; ++w) {
			if (fin[w]) { a.seeds[sid[w]].rbeg = (int64_t)(steps[w] + sa_v[w]); have[w] = false; ++n_sa; }
			if (walk[w]) { n_lf += k[w] != ix.primary; k[w] = lane_lf_count(ix, k[w], blk[w]); ++steps[w]; }
		}
	}
	unsigned long long lf = n_lf, sa = n_sa;
	for (int m = 32; m; m >>= 1) { lf += __shfl_xor(lf, m); sa += __shfl_xor(sa, m); }
	if (lane_id() == 0 && sa) { atomicAdd(&cnt_row(a.counters)[CNT_SA], sa); atomicAdd(&cnt_row(a.counters)[CNT_LF], lf); atomicAdd(&cnt_row(a.counters)[CNT_SEEDS], sa); }
}

// K2c -- contig id of every seed (bns_intv2rid, bntseq.c:370, as called at bwamem.c:293)
__global__ __launch_bounds__(256) void k_seed_rid(SeedLaunch a, long long total)
{
	const long long sid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (sid >= total) return;
	const int64_t rbeg = a.seeds[sid].rbeg;
	a.seeds[sid].rid = dev_intv2rid(a.ix, rbeg, rbeg + a.seeds[sid].len);
}

// ------------------------------------------------------------- the bases of the Occ blocks as bit planes (fmi_dev.h: count_bases64)
__device__ __forceinline__ uint32_t even_bits16(uint32_t x)   // bits 30, 28, .. 0 of x -> bits 15 .. 0
{
	x &= 0x55555555u;
	x = (x | x >> 1) & 0x33333333u; x = (x | x >> 2) & 0x0f0f0f0fu; x = (x | x >> 4) & 0x00ff00ffu; x = (x | x >> 8) & 0x0000ffffu;
	return x;
}
// words 8..15 of every 16-word block: eight words of sixteen 2-bit codes (first base on top, bwt.h:80) -> {H0,H1,L0,L1, H2,H3,L2,L3}
__global__ __launch_bounds__(256) void k_bwt_planes(uint32_t *bwt, uint64_t n_blocks)
{
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
		uint4 *p = reinterpret_cast<uint4*>(bwt + b * 16 + 8);
		const uint4 lo = p[0], hi = p[1];
		const uint32_t w[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
		uint32_t H[4], L[4];
		for (int g = 0; g < 4; ++g) {
			H[g] = even_bits16(w[2 * g] >> 1) << 16 | even_bits16(w[2 * g + 1] >> 1);
			L[g] = even_bits16(w[2 * g]) << 16 | even_bits16(w[2 * g + 1]);
		}
		p[0] = make_uint4(H[0], H[1], L[0], L[1]);
		p[1] = make_uint4(H[2], H[3], L[2], L[3]);
	}
}

// ------------------------------------------------------------- the suffix array, denser than the index files hold it
// bwa index keeps SA[k] for every 32nd BWT row k and bwt_sa (bwt.c:86) walks LF steps from a row to the next sampled one: 31 dependent 64-byte
// gathers per look-up on average (a step lands on a sampled row with probability 1/32), 290 per read -- 8 ms per 1 M-read batch, a quarter of
// the BWT search itself.  With 288 GB of HBM the table can hold every row: k_sa_fill computes the rows in between ON THE GPU when the index is
// loaded (SA[k] = SA[LF(k)] + 1, the very sum bwt_sa forms, so every look-up returns the value the reference computes), halving the interval
// pass by pass; a pass resolves N / I_from rows in I_from steps each, i.e. one LF step per BWT row and pass.  The files stay bwa's.
__global__ __launch_bounds__(256) void k_sa_spread(const uint64_t *sa, uint64_t n_sa, int shift, uint64_t *dense)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sa; i += (uint64_t)gridDim.x * blockDim.x) dense[i << shift] = sa[i];
}
// rows k = (2j + 1) * to_intv, j < n_new: walk to a row that is a multiple of 2 * to_intv (filled by the earlier passes)
__global__ __launch_bounds__(256) void k_sa_fill(DevIndex ix, uint64_t *dense, int dense_shift, uint64_t to_intv, uint64_t n_new)
{
	const uint64_t n_lanes = (uint64_t)gridDim.x * blockDim.x, mask = 2 * to_intv - 1;
	uint64_t next = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t k0 = 0, k = 0, steps = 0;
	bool have = false;
	for (;;) {
		if (!have && next < n_new) { k0 = k = (2 * next + 1) * to_intv; steps = 0; have = true; next += n_lanes; }
		if (__ballot(have) == 0) break;
		if (have && (k & mask) == 0) { dense[k0 >> dense_shift] = steps + dense[k >> dense_shift]; have = false; }
		else if (have) { k = lane_lf(ix, k); ++steps; }
	}
}

// ------------------------------------------------------------- known-answer kernels
__global__ void k_kat_occ4(DevIndex ix, int n, const uint64_t *k, uint64_t *out)
{
	int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
	bool live = q < n;
	uint64_t cnt[4];
	quad_occ4(ix, live ? k[q] : 0, live, cnt);
	if (live && (lane_id() & 3) == 0) { out[q*4+0] = cnt[0]; out[q*4+1] = cnt[1]; out[q*4+2] = cnt[2]; out[q*4+3] = cnt[3]; }
}
__global__ void k_kat_sa(DevIndex ix, int n, const uint64_t *k, uint64_t *out)
{
	int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
	bool live = q < n;
	unsigned long long lf = 0;
	uint64_t v = quad_sa(ix, live ? k[q] : 0, live, lf);
	if (live && (lane_id() & 3) == 0) out[q] = v;
}
__global__ void k_kat_extend(DevIndex ix, int n, const uint64_t *ik3, const int *is_back, uint64_t *ok12)
{
	int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	bool live = g < n;
	Bi ik = { 0, 0, 0 }, ok[4];
	int back = 0;
	if (live) { ik.x0 = ik3[g*3]; ik.x1 = ik3[g*3+1]; ik.x2 = ik3[g*3+2]; back = is_back[g]; }
	group8_extend(ix, ik, back, live, ok);
	if (live && (lane_id() & 7) == 0)
		for (int i = 0; i < 4; ++i) { ok12[g*12 + i*3] = ok[i].x0; ok12[g*12 + i*3 + 1] = ok[i].x1; ok12[g*12 + i*3 + 2] = ok[i].x2; }
}

} // namespace

int launch_seeds(const SeedLaunch &a, int64_t total, hipStream_t st)
{
	if (total <= 0) return 0;
	hipLaunchKernelGGL(k_seed_rows, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, a);
	long long blocks = (total + 255) / 256;
	if (blocks > 256 * 8) blocks = 256 * 8;                   // 8 waves per SIMD resident; lanes stride over the rest
	static const int walks = getenv("BWAHIP_SEED_WALKS") ? atoi(getenv("BWAHIP_SEED_WALKS")) : 1;
	if (walks >= 2) hipLaunchKernelGGL(k_seed_walk<2>, dim3((unsigned)blocks), dim3(256), 0, st, a, (long long)total);
	else hipLaunchKernelGGL(k_seed_walk<1>, dim3((unsigned)blocks), dim3(256), 0, st, a, (long long)total);
	hipLaunchKernelGGL(k_seed_rid, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, (long long)total);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_bwt_planes(uint32_t *bwt, uint64_t n_words, hipStream_t st)
{
	const uint64_t n_blocks = n_words / 16;
	if (n_blocks) hipLaunchKernelGGL(k_bwt_planes, dim3(256 * 16), dim3(256), 0, st, bwt, n_blocks);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
// dense must hold (seq_len >> log2(to_intv)) + 1 entries; ix.sa / ix.sa_intv are the table of the index files
int launch_sa_densify(const DevIndex &ix, uint64_t *dense, int to_intv, hipStream_t st)
{
	int dense_shift = 0, spread = 0;
	while ((1 << dense_shift) < to_intv) ++dense_shift;
	while ((to_intv << spread) < ix.sa_intv) ++spread;
	if ((1 << dense_shift) != to_intv || (to_intv << spread) != ix.sa_intv) return BWAHIP_EINVAL;
	hipLaunchKernelGGL(k_sa_spread, dim3(256 * 8), dim3(256), 0, st, ix.sa, ix.n_sa, spread, dense);
	for (uint64_t I = (uint64_t)ix.sa_intv >> 1; I >= (uint64_t)to_intv; I >>= 1) {
		const uint64_t n_new = (ix.seq_len / I + 1) / 2;       // odd multiples of I up to seq_len
		if (n_new) hipLaunchKernelGGL(k_sa_fill, dim3(256 * 8), dim3(256), 0, st, ix, dense, dense_shift, I, n_new);
	}
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_kat_occ4(const DevIndex &ix, int n, const uint64_t *k, uint64_t *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_kat_occ4, dim3((n * 4 + 255) / 256), dim3(256), 0, st, ix, n, k, out);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_kat_sa(const DevIndex &ix, int n, const uint64_t *k, uint64_t *out, hipStream_t st)
{
	hipLaunchKernelGGL(k_kat_sa, dim3((n * 4 + 255) / 256), dim3(256), 0, st, ix, n, k, out);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_kat_extend(const DevIndex &ix, int n, const uint64_t *ik3, const int *is_back, uint64_t *ok12, hipStream_t st)
{
	hipLaunchKernelGGL(k_kat_extend, dim3((n * 8 + 255) / 256), dim3(256), 0, st, ix, n, ik3, is_back, ok12);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
