// K3b -- mem_flt_chained_seeds (bwamem.c:605-622) with mem_seed_sw (bwamem.c:578-603): for reads long enough
// relative to the minimum chain weight (-W) or to 5.5*ln(l), every seed of every kept chain shorter than 200 bases is
// re-scored by a local alignment of the seed +/- 50 bases (the reference's SSE2 ksw_align2, word kernel) and dropped
// when the score stays below min_HSP_score; surviving seeds carry the SW score into mem_chain2aln's seed order.
// One read per wavefront; its seeds are spread over the 8 eight-lane groups of ssw_dev.h (one alignment per group),
// then one lane per chain compacts the chain's seed list in place.  Integer DP: MFMA not applicable.
// Also the known-answer kernel for ksw_align2 (byte and word kernels) the parity tests drive.
#include "bwahip_internal.h"
#include "ssw_dev.h"

namespace {

constexpr int SHORT_EXT = 50, SHORT_LEN = 200;               // MEM_SHORT_EXT / MEM_SHORT_LEN, bwamem.c:570-571
constexpr int GROUPS = 8, GP = 8;                            // word kernel: 8 lanes per alignment
constexpr int SEG = (SHORT_LEN + GP - 1) / GP * GP;          // cells per H/E array of one group

__device__ __forceinline__ int pos2rid(const DevIndex &ix, int64_t pos_f)   // bntseq.c:354
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
__device__ __forceinline__ int ref_base(const DevIndex &ix, int64_t p) { return p < ix.l_pac ? pac_at(ix.pac, p) : 3 - pac_at(ix.pac, (ix.l_pac << 1) - 1 - p); }

__global__ __launch_bounds__(64) void k_seed_sw(SeedSwLaunch a)
{
	__shared__ uint8_t s_q[BWAHIP_MAX_READ_LEN + 8];
	__shared__ int8_t s_mat[32];
	__shared__ uint8_t s_t[GROUPS][SHORT_LEN + 8];
	__shared__ int8_t s_prof[GROUPS][5 * SEG];
	__shared__ int16_t s_h[GROUPS][3 * SEG];
	const int r = blockIdx.x, lane = (int)(threadIdx.x & 63), g = lane >> 3, gl = lane & 7;
	const DevOpt &opt = a.opt;
	const DevIndex &ix = a.ix;
	const int l_query = (int)(a.off[r + 1] - a.off[r]);
	const int n_chn = a.chain_n[r];
	if (n_chn <= 0 || l_query <= 0) return;
	// bwamem.c:607-609 (types as in the reference: float product for -W, double for the log form; the test in double)
	const double min_l = opt.min_chain_weight ? (double)(1.1f * (float)opt.min_chain_weight) : (double)5.5f * a.logtab[l_query];
	const int min_hsp = (int)(opt.a * min_l + .499);
	if (min_l > (double)(0.05f * (float)l_query)) return;
	const int64_t sb = a.seed_base[r], l_pac = ix.l_pac;
	const int tot = a.kept_seeds[r];
	const uint8_t *query = a.seq + a.off[r];
	for (int i = lane; i < l_query; i += 64) s_q[i] = query[i];
	if (lane < 25) s_mat[lane] = opt.mat[lane];
	__syncthreads();
	ssw::Work w = { s_prof[g], s_h[g], s_h[g] + SEG, s_h[g] + 2 * SEG, nullptr, nullptr };
	DevSeed *seeds = a.chain_seeds + sb;
	for (int t0 = 0; t0 < tot; t0 += GROUPS) {
		const int t = t0 + g;
		if (t < tot) {
			const DevSeed s = seeds[t];
			int sc = -1;
			if (s.len < SHORT_LEN) {                                // bwamem.c:585
				int qb = s.qbeg - SHORT_EXT, qe = s.qbeg + s.len + SHORT_EXT;
				int64_t rb = s.rbeg - SHORT_EXT, re = s.rbeg + s.len + SHORT_EXT;
				const int64_t mid = (s.rbeg + s.rbeg + s.len) >> 1;
				qb = qb > 0 ? qb : 0; qe = qe < l_query ? qe : l_query;
				rb = rb > 0 ? rb : 0; re = re < l_pac << 1 ? re : l_pac << 1;
				if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
				if (!(qe - qb >= SHORT_LEN || re - rb >= SHORT_LEN)) {
					// bns_fetch_seq (bntseq.c:426): clamp to the contig that holds `mid`
					const bool is_rev = mid >= l_pac;
					const int rid = pos2rid(ix, is_rev ? (l_pac << 1) - 1 - mid : mid);
					int64_t far_beg = ix.anns[rid].offset, far_end = far_beg + ix.anns[rid].len;
					if (is_rev) { const int64_t tmp = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - tmp; }
					rb = rb > far_beg ? rb : far_beg; re = re < far_end ? re : far_end;
					const int tlen = (int)(re - rb), qlen = qe - qb;
					for (int i = gl; i < tlen; i += GP) s_t[g][i] = (uint8_t)ref_base(ix, rb + i);
					int slen, shift, qmax;
					const ssw::SeqView qv = { s_q + qb, 1, 0 }, tv = { s_t[g], 1, 0 };
					ssw::qinit<GP>(w, gl, qlen, qv, s_mat, slen, shift, qmax);
					__threadfence_block();
					// KSW_XSTART only: no sub-optimal list, no early stop; only the score of the first pass is used (bwamem.c:601-603)
					const ssw::Res x = ssw::pass<GP, false>(w, lane, slen, shift, qmax, tlen, tv, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, 0x10000, 0x10000);
					sc = x.score;
				}
			}
			if (gl == 0) seeds[t].score = sc;
		}
	}
	__threadfence_block();
	__syncthreads();
	// bwamem.c:612-620: per chain, keep the seeds without a score (-1: taken as is) or with score >= min_HSP_score
	for (int ci = lane; ci < n_chn; ci += 64) {
		DevChain *c = a.chains + sb + ci;
		DevSeed *cs = seeds + c->seed_off;
		int k = 0;
		for (int j = 0; j < c->n; ++j) {
			DevSeed s = cs[j];
			if (s.score < 0 || s.score >= min_hsp) {
				s.score = s.score < 0 ? s.len * opt.a : s.score;
				cs[k++] = s;
			}
		}
		c->n = k;
	}
}

// known-answer kernel: ksw_align2 on caller-supplied pairs.  params per item: qlen, tlen, xtra, o_del, e_del, o_ins, e_ins, 0.
// One item per group; byte kernel when xtra has KSW_XBYTE.  Working set in global memory (any query length).
template <int P>
__device__ void kat_item(const int8_t *mat, const int *p, const uint8_t *q, const uint8_t *t, uint8_t *wsp, int lane, int *out7)
{
	const int qlen = p[0], tlen = p[1];
	const int cells = (qlen + P - 1) / P * P;
	ssw::Work w;
	w.prof = (int8_t*)wsp;
	w.H0 = (int16_t*)(wsp + 5 * (size_t)cells + 8 - (5 * (size_t)cells) % 8);
	w.H1 = w.H0 + cells; w.E = w.H1 + cells; w.Hmax = w.E + cells;
	w.colmax = (uint16_t*)(w.Hmax + cells);
	// queries of up to 16 segments take the register-resident variant (as k_matesw does), longer ones the LDS/global one
	const ssw::Res r = ssw::align2<P, 16>(w, lane, qlen, q, 1, tlen, t, 1, mat, p[3], p[4], p[5], p[6], p[2]);
	if ((lane & (P - 1)) == 0) { out7[0] = r.score; out7[1] = r.te; out7[2] = r.qe; out7[3] = r.score2; out7[4] = r.te2; out7[5] = r.tb; out7[6] = r.qb; }
}
__global__ __launch_bounds__(64) void k_kat_align(DevOpt opt, int n, int byte_mode, const int *items, const int *params, const uint8_t *q, const int64_t *qoff,
                                                  const uint8_t *t, const int64_t *toff, uint8_t *wsp, size_t wsp_stride, int *out7)
{
	__shared__ int8_t s_mat[32];
	const int lane = (int)(threadIdx.x & 63);
	if (lane < 25) s_mat[lane] = opt.mat[lane];
	__syncthreads();
	const int P = byte_mode ? 16 : 8, per = 64 / P;
	const int slot = blockIdx.x * per + lane / P;
	if (slot >= n) return;
	const int it = items[slot];
	uint8_t *my = wsp + (size_t)slot * wsp_stride;
	if (byte_mode) kat_item<16>(s_mat, params + 8 * it, q + qoff[it], t + toff[it], my, lane, out7 + 7 * it);
	else kat_item<8>(s_mat, params + 8 * it, q + qoff[it], t + toff[it], my, lane, out7 + 7 * it);
}

} // namespace

int launch_seed_sw(const SeedSwLaunch &a, hipStream_t st)
{
	if (a.n_reads <= 0) return 0;
	hipLaunchKernelGGL(k_seed_sw, dim3(a.n_reads), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_kat_align(const DevOpt &opt, int n, int byte_mode, const int *items, const int *params, const uint8_t *q, const int64_t *qoff,
                     const uint8_t *t, const int64_t *toff, uint8_t *wsp, size_t wsp_stride, int *out7, hipStream_t st)
{
	if (n <= 0) return 0;
	const int per = byte_mode ? 4 : 8;
	hipLaunchKernelGGL(k_kat_align, dim3((n + per - 1) / per), dim3(64), 0, st, opt, n, byte_mode, items, params, q, qoff, t, toff, wsp, wsp_stride, out7);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
