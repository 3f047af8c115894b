// K2 -- seed positions: suffix-array look-up for every occurrence chaining will use
// (the `bwt_sa` call of mem_chain, bwamem.c:290, with the max_occ striding of bwamem.c:285-286),
// plus the contig id test of bwamem.c:293 (bns_intv2rid, bntseq.c:370).
//
// All look-ups of a batch are independent, so they are flattened: an exclusive scan of the per-read
// look-up counts (written by K1) gives every seed a fixed slot, in exactly the order the reference
// visits them (interval order, then k), and one quad of lanes resolves one seed: up to sa_intv-1
// dependent LF steps (bwt.c:53), each one fully used 64-byte gather, then one 8-byte SA read.
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

__global__ __launch_bounds__(256) void k_seeds(SeedLaunch a, long long total)
{
	const long long quad0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
	const long long n_quads = ((long long)gridDim.x * blockDim.x) >> 2;
	const int r4 = lane_id() & 3;
	unsigned long long n_lf = 0, n_sa = 0;
	const long long rounds = (total + n_quads - 1) / n_quads;
	for (long long it = 0; it < rounds; ++it) {              // same trip count for every lane: the DPP steps stay convergent
		long long sid = quad0 + it * n_quads;
		bool live = sid < total;
		uint64_t row = 0; int qbeg = 0, slen = 0;
		if (live) {
			int lo = 0, hi = a.n_reads;                      // largest r with seed_base[r] <= sid
			while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.seed_base[mid] <= sid) lo = mid; else hi = mid; }
			int r = lo, o = (int)(sid - a.seed_base[r]), n = a.intv_n[r];
			const DevIntv *iv = a.intv + (size_t)r * a.cap;
			for (int t = 0; t < n; ++t) {
				uint64_t x2 = iv[t].x2;
				uint64_t step = x2 > (uint64_t)a.opt.max_occ ? x2 / a.opt.max_occ : 1;
				uint64_t cnt = (x2 + step - 1) / step;
				if (cnt > (uint64_t)a.opt.max_occ) cnt = a.opt.max_occ;
				if ((uint64_t)o < cnt) {
					uint64_t info = iv[t].info;
					row = iv[t].x0 + (uint64_t)o * step;
					qbeg = (int)(info >> 32); slen = (int)(uint32_t)info - qbeg;
					break;
				}
				o -= (int)cnt;
			}
		}
		uint64_t rbeg = quad_sa(a.ix, row, live, n_lf);
		if (live && r4 == 0) {
			DevSeed sd;
			sd.rbeg = (int64_t)rbeg; sd.qbeg = qbeg; sd.len = slen; sd.score = slen;
			sd.rid = dev_intv2rid(a.ix, sd.rbeg, sd.rbeg + slen);
			a.seeds[sid] = sd;
			++n_sa;
		}
	}
	if (r4 == 0 && n_sa) { atomicAdd(&a.counters[CNT_SA], n_sa); atomicAdd(&a.counters[CNT_LF], n_lf); atomicAdd(&a.counters[CNT_SEEDS], n_sa); }
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
	long long quads = total;
	long long blocks = (quads * 4 + 255) / 256;
	if (blocks > 256 * 32) blocks = 256 * 32;                 // grid-stride the rest
	hipLaunchKernelGGL(k_seeds, dim3((unsigned)blocks), dim3(256), 0, st, a, (long long)total);
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
