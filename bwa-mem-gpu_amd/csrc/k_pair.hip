// Paired-end stages on the GPU (the reference's mem_pestat + mem_sam_pe, bwamem_pair.c):
//   k_pestat    per pair: the insert size mem_pestat would collect (bwamem_pair.c:72-97), as a histogram; the host turns
//               the histogram into mem_pestat_t (percentiles, mean, sd: a few thousand numbers) and into the table of
//               .721*log(2*erfc(|ns|/sqrt 2)) per distance that mem_pair needs (glibc's erfc / log, bwamem_pair.c:244).
//   k_pe_prepare / k_pe_copy   capacities of the per-read lists after rescue; the pairs that need a Smith-Waterman.
//   k_matesw    mem_matesw (bwamem_pair.c:137-206) for those pairs, one pair per wavefront, in the reference's order
//               (the skip test looks at the list as rescued so far); ksw_align2 by the lane-exact striped SW of
//               ssw_dev.h (byte kernel for reads under 250 bases); mem_sort_dedup_patch without patching afterwards.
//   k_pair      mem_pair (bwamem_pair.c:208-272) and the decisions of mem_sam_pe (bwamem_pair.c:303-365, 397-411): which
//               regions print, with which mapQ and flags.  Sorting inside mem_pair is on keys that cannot tie, so a
//               rank sort gives the reference's order; of the list u only its top two elements and a count are needed.
// Integer / table work throughout: MFMA not applicable.
#include "bwahip_internal.h"
#include "wave_dev.h"
#include "regsort_dev.h"
#include "ssw_dev.h"
#include "final_dev.h"

namespace {
using namespace wv;
using namespace fin;

__device__ __forceinline__ uint64_t hash_64(uint64_t key)      // utils.h:97
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

__device__ __forceinline__ int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)   // bwamem_pair.c:48
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

__device__ int cal_sub(const DevOpt &o, const DevReg *a, int n)   // bwamem_pair.c:58-70
{
	int j;
	for (j = 1; j < n; ++j) {
		const int b_max = a[j].qb > a[0].qb ? a[j].qb : a[0].qb;
		const int e_min = a[j].qe < a[0].qe ? a[j].qe : a[0].qe;
		if (e_min > b_max) {
			const int min_l = a[j].qe - a[j].qb < a[0].qe - a[0].qb ? a[j].qe - a[j].qb : a[0].qe - a[0].qb;
			if ((float)(e_min - b_max) >= (float)min_l * o.mask_level) break;
		}
	}
	return j < n ? a[j].score : o.min_seed_len * o.a;
}

// ---------------------------------------------------------------------------------------------------- mem_pestat, device part
__global__ void k_pestat(PairLaunch a)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= a.n_reads >> 1) return;
	const int r0 = p << 1, r1 = r0 | 1;
	const int n0 = a.reg_n[r0], n1 = a.reg_n[r1];
	if (n0 == 0 || n1 == 0) return;
	const DevReg *a0 = a.regs + a.reg_base[r0], *a1 = a.regs + a.reg_base[r1];
	if ((double)cal_sub(a.opt, a0, n0) > 0.8 * a0[0].score) return;
	if ((double)cal_sub(a.opt, a1, n1) > 0.8 * a1[0].score) return;
	if (a0[0].rid != a1[0].rid) return;
	int64_t is;
	const int dir = infer_dir(a.ix.l_pac, a0[0].rb, a1[0].rb, &is);
	if (is && is <= a.opt.max_ins) atomicAdd(&a.hist[(size_t)dir * (a.opt.max_ins + 1) + is], 1u);
}

// ---------------------------------------------------------------------------------------------------- rescue: preparation
// per pair: nb[] (bwamem_pair.c:291-297: regions within pen_unpaired of the best one, at most max_matesw are used) and the
// capacity of each list after rescue: every used region of the mate can add one region per live orientation
__global__ void k_pe_prepare(PairLaunch a)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= a.n_reads >> 1) return;
	int n_live = 0;
	for (int d = 0; d < 4; ++d) n_live += a.pes[d].failed ? 0 : 1;
	int nb[2];
	for (int i = 0; i < 2; ++i) {
		const int r = p << 1 | i, n = a.reg_n[r];
		const DevReg *l = a.regs + a.reg_base[r];
		int c = 0;
		if (!(a.opt.flag & BWAHIP_F_NO_RESCUE) && n > 0) {
			const int thr = l[0].score - a.opt.pen_unpaired;
			for (int j = 0; j < n; ++j) c += l[j].score >= thr ? 1 : 0;   // the list is sorted by score: a prefix
			c = c < a.opt.max_matesw ? c : a.opt.max_matesw;
		}
		nb[i] = c; a.nb[r] = c; a.sw_cnt[r] = 4 * c;
	}
	a.pe_cap[p << 1] = a.reg_n[p << 1] + nb[1] * n_live;
	a.pe_cap[p << 1 | 1] = a.reg_n[p << 1 | 1] + nb[0] * n_live;
}

// the reference window of mem_matesw for anchor `an`, orientation r and a mate of l_ms bases (bwamem_pair.c:153-166);
// true when the alignment is attempted (same contig as the anchor, window at least one seed long)
__device__ __forceinline__ bool ms_window(const PairLaunch &a, const DevReg &an, int r, int l_ms, int64_t &rb, int64_t &re)
{
	const DevIndex &ix = a.ix;
	const int64_t l_pac = ix.l_pac;
	const int is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
	if (!is_rev) {
		rb = is_larger ? an.rb + a.pes[r].low : an.rb - a.pes[r].high;
		re = (is_larger ? an.rb + a.pes[r].high : an.rb - a.pes[r].low) + l_ms;
	} else {
		rb = (is_larger ? an.rb + a.pes[r].low : an.rb - a.pes[r].high) - l_ms;
		re = is_larger ? an.rb + a.pes[r].high : an.rb - a.pes[r].low;
	}
	if (rb < 0) rb = 0;
	if (re > l_pac << 1) re = l_pac << 1;
	int rid = -1;
	if (rb < re) {                                                // bns_fetch_seq (bntseq.c:426): clamp to the contig holding the middle
		const int64_t mid = (rb + re) >> 1;
		const bool mrev = mid >= l_pac;
		rid = dev_pos2rid(ix, mrev ? (l_pac << 1) - 1 - mid : mid);
		int64_t far_beg = ix.anns[rid].offset, far_end = far_beg + ix.anns[rid].len;
		if (mrev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
		rb = rb > far_beg ? rb : far_beg;
		re = re < far_end ? re : far_end;
	}
	return an.rid == rid && re - rb >= a.opt.min_seed_len;
}

constexpr int SW_TW = 1024;                                  // widest window k_matesw_sw takes (wider ones are aligned inside k_matesw)
constexpr int SW8_QMAX = 256;                                // longest mate the word kernel's alignments are run ahead for (32 cells per lane in registers)

// per pair: copy both lists into their (larger) slots; list the pair for k_matesw if any anchor has an orientation left.
// W lanes work on the pair (1: a thread of k_pe_copy; 64: a wavefront of k_pe_copy_big, for the pairs with more than PE_COPY_SMALL regions --
// 50 anchors x a list of thousands is 10^5 steps in a row for one thread, and a human-like batch has tens of thousands of such pairs)
constexpr int PE_COPY_SMALL = 32;
template <int W>
__device__ __forceinline__ void pe_copy_pair(const PairLaunch &a, int p, int l)
{
	bool need = false;
	int failed = 0;
	for (int d = 0; d < 4; ++d) failed |= a.pes[d].failed ? 1 << d : 0;
	for (int i = 0; i < 2; ++i) {
		const int r = p << 1 | i, n = a.reg_n[r];
		const DevReg *src = a.regs + a.reg_base[r];
		DevReg *dst = a.pe_regs + a.pe_base[r];
		for (int j = l; j < n; j += W) dst[j] = src[j];
		if (l == 0) a.pe_n[r] = n;
		const int m = r ^ 1;
		const DevReg *ma = a.regs + a.reg_base[m];
		const int n_m = a.reg_n[m];
		const int l_ms = (int)(a.off[m + 1] - a.off[m]);
		for (int j = 0; j < a.nb[r]; ++j) {
			// skip mask of mem_matesw (bwamem_pair.c:143-150): the orientations in which the mate already has a hit at a proper distance
			const int64_t arb = src[j].rb;
			int sk = failed;
			for (int k = l; k < n_m; k += W) {
				int64_t dist;
				const int d = infer_dir(a.ix.l_pac, arb, ma[k].rb, &dist);
				if (dist >= a.pes[d].low && dist <= a.pes[d].high) sk |= 1 << d;
			}
			if (W > 1) sk = (__ballot(sk & 1) ? 1 : 0) | (__ballot(sk & 2) ? 2 : 0) | (__ballot(sk & 4) ? 4 : 0) | (__ballot(sk & 8) ? 8 : 0);
			if (sk == 15) continue;
			need = true;
			// the alignments this anchor asks for against the unrescued list can be done ahead, in parallel (k_matesw_sw)
			// (mates of l_ms * a < 250: the byte kernel, k_matesw_sw; up to 256 bases beyond that: the word kernel, k_matesw_sw8, its tasks from the
			// end of the same array; longer mates are aligned inside k_matesw)
			const bool byte_k = l_ms * a.opt.a < 250;
			if (byte_k || l_ms <= SW8_QMAX)
				for (int o = W > 1 ? l : 0; o < 4; o += W > 1 ? 64 : 1) {
					int64_t rb, re;
					if ((sk >> o & 1) || !ms_window(a, src[j], o, l_ms, rb, re) || re - rb > SW_TW) continue;
					const int slot = (int)a.sw_base[r] + 4 * j + o;
					a.sw_res[slot].state = 1;
					if (byte_k) a.sw_tasks[atomicAdd(a.sw_n, 1)] = slot;
					else a.sw_tasks[a.sw_cap - 1 - atomicAdd(a.sw_n8, 1)] = slot;
					a.sw_info[slot] = make_int2(r, j << 2 | o);
				}
		}
	}
	if (need && l == 0) { a.resc_list[atomicAdd(a.resc_n, 1)] = p; a.resc_flag[p] = 1; }
}
__global__ void k_pe_copy(PairLaunch a)
{
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= a.n_reads >> 1) return;
	if (a.reg_n[p << 1] + a.reg_n[p << 1 | 1] > PE_COPY_SMALL) return;     // k_pe_copy_big's
	pe_copy_pair<1>(a, p, 0);
}
__global__ __launch_bounds__(256) void k_pe_copy_big(PairLaunch a)
{
	const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (p >= a.n_reads >> 1) return;
	if (a.reg_n[p << 1] + a.reg_n[p << 1 | 1] <= PE_COPY_SMALL) return;
	pe_copy_pair<64>(a, p, lane());
}

// ---------------------------------------------------------------------------------------------------- mem_matesw
constexpr int MS_MAXQ = BWAHIP_MAX_READ_LEN;
constexpr int MS_LIST = 224;                                 // regions of a mate list worked on in LDS

struct MsLds {
	uint8_t *q;                                                  // mate read (or its reverse complement), codes
	int8_t *mat;
	int8_t *prof; int16_t *h;                                    // striped-SW working set: 5 + 4 x 2 bytes per cell
	RegKey *keys; int *idx; int *stk;                            // sort scratch
	uint8_t *tw; int tw_cap;                                     // reference window in LDS when it fits (a dependent global load per DP column costs ~1 us)
	uint16_t *cm;                                                // column maxima (tw_cap entries) in LDS for the same reason
};

// A region list the helpers below may index with offsets the compiler folds into the instruction (&L[i - 1] as base, +80 as offset): with a
// generic pointer the hardware picks LDS or global from the BASE register alone, and a base one record in front of an LDS array lies
// outside the LDS aperture -- a memory violation (round 2).  The helpers therefore take a ListRef, and a ListRef into LDS can only be
// made from a GuardedLds, whose first and last records are never handed out: whatever offset within one record gets folded, the base stays in LDS.
template <int N> struct __attribute__((aligned(16))) GuardedLds { DevReg g[N + 2]; };
class ListRef {
	DevReg *p_;
	explicit __device__ ListRef(DevReg *p) : p_(p) {}
public:
	template <int N> static __device__ __forceinline__ ListRef lds(GuardedLds<N> &a) { return ListRef(a.g + 1); }
	static __device__ __forceinline__ ListRef global(DevReg *q) { return ListRef(q); }
	__device__ __forceinline__ DevReg &operator[](int i) const { return p_[i]; }
	__device__ __forceinline__ DevReg *ptr() const { return p_; }
};

// mem_sort_dedup_patch with bns == 0 (no patching), as mem_matesw calls it (bwamem_pair.c:203; bwamem.c:444-496).
// L: the list (n entries), tmp: a spare list of the same capacity, keys / idx: sort scratch (n and 2n entries).
// re_ties: two entries of the list shared their `re` (the first sort's order between them, and with it the outcome of their redundancy
// test, then depends on the arrangement the list came in: see incr_insert)
__device__ __forceinline__ int sort_dedup_nopatch(const DevOpt &o, int n, const ListRef L, const ListRef tmp, RegKey *keys, int *idx, int *stk, unsigned *lds256, int l, int *err, bool &re_ties)
{
	re_ties = false;
	if (n <= 1) return n;
	for (int i = l; i < n; i += 64) { keys[i].k64 = L[i].re; keys[i].score = 0; keys[i].qb = 0; idx[i] = i; }
	wsync();
	// the whole wavefront sorts, exactly also when keys are equal (a rescued hit that repeats one of the list: the usual case here); its
	// scratch is the spare list, which is free while a sort runs
	if (!wave_sort_exact(RegSort{keys, 0}, n, idx, reinterpret_cast<int*>(tmp.ptr()), stk, lds256, l)) {
		wsync();
		if (l == 0) { int bad = 0; rs_introsort(RegSort{keys, 0}, n, idx, stk, &bad); if (bad) atomicExch(err, 40 + bad); }
	}
	wsync();
	for (int i = l; i < n; i += 64) { tmp[i] = L[idx[i]]; tmp[i].n_comp = 1; }
	wsync();
	for (int i = l; i < n; i += 64) L[i] = tmp[i];
	wsync();
	for (int base = 0; base < n; base += 64) { const int i = base + l; if (__ballot(i >= 1 && i < n && L[i].re == L[i - 1].re)) re_ties = true; }
	// redundancy (bwamem.c:451-480 without the patch branch).  The test that lets an entry take part (bwamem.c:453) reads rid, rb and re, which
	// nothing here changes: it is evaluated 64 entries at a time, and lane 0 then visits the entries that passed
	for (int cbase = 0; cbase < n; cbase += 64) {
		const int ii = cbase + l;
		unsigned long long act_m = __ballot(ii >= 1 && ii < n && L[ii].rid == L[ii - 1].rid && L[ii].rb < L[ii - 1].re + o.max_chain_gap);
		if (l == 0) while (act_m) {
			const int i = cbase + __ffsll((long long)act_m) - 1;
			act_m &= act_m - 1;
			DevReg *p = &L[i];
			for (int j = i - 1; j >= 0 && p->rid == L[j].rid && p->rb < L[j].re + o.max_chain_gap; --j) {
				DevReg *q = &L[j];
				if (q->qe == q->qb) continue;
				const int64_t orr = q->re - p->rb, oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
				const int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
				const int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
				if ((float)orr > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {
					if (p->score < q->score) { p->qe = p->qb; break; }
					else q->qe = q->qb;
				}
			}
		}
	}
	wsync();
	int m = 0;
	for (int base = 0; base < n; base += 64) {                    // drop the excluded ones, order kept
		const int i = base + l;
		const bool keep = i < n && L[i].qe > L[i].qb;
		const unsigned long long km = __ballot(keep);
		if (keep) tmp[m + __popcll(km & ((1ull << l) - 1))] = L[i];
		m += __popcll(km);
	}
	wsync();
	n = m;
	for (int i = l; i < n; i += 64) { L[i] = tmp[i]; keys[i].k64 = tmp[i].rb; keys[i].score = tmp[i].score; keys[i].qb = tmp[i].qb; idx[i] = i; }
	wsync();
	if (!wave_sort_exact(RegSort{keys, 1}, n, idx, reinterpret_cast<int*>(tmp.ptr()), stk, lds256, l)) {
		wsync();
		if (l == 0) { int bad = 0; rs_introsort(RegSort{keys, 1}, n, idx, stk, &bad); if (bad) atomicExch(err, 50 + bad); }
	}
	wsync();
	for (int i = l; i < n; i += 64) tmp[i] = L[idx[i]];
	wsync();
	m = 0;
	for (int base = 0; base < n; base += 64) {                    // identical hits (bwamem.c:488-494)
		const int i = base + l;
		bool keep = i < n;
		if (i > 0 && i < n) keep = !(tmp[i].score == tmp[i-1].score && tmp[i].rb == tmp[i-1].rb && tmp[i].qb == tmp[i-1].qb);
		const unsigned long long km = __ballot(keep);
		if (keep) L[m + __popcll(km & ((1ull << l) - 1))] = tmp[i];
		m += __popcll(km);
	}
	wsync();
	return m;
}

// One region into a list that the last mem_sort_dedup_patch left -- WITHOUT sorting it again (round 3).  mem_matesw re-runs
// mem_sort_dedup_patch (bns == 0) on the whole list after every rescued hit (bwamem_pair.c:203); inside a repeat family that is a list of
// thousands of regions, a hundred times per pair.  When no two entries share `re` and no two share (score, rb, qb), both sorts have exactly one
// possible result whatever arrangement they start from, and the call is a function of the SET of regions; and a list that such a call returned
// is a fixed point of it: every pair of survivors within max_chain_gap of each other went through the redundancy test (bwamem.c:459-470: the
// later one's loop only ends early when that one is excluded) and passed.  So for L + {b} the only tests with a new outcome are b's:
//   * b's own turn: the entries before it in `re` order and within reach (bwamem.c:454,456), nearest first; a redundant pair excludes the one
//     of lower score -- b, which ends its turn, or the other (q on a tie), and b goes on;
//   * then the turn of every entry after b in `re` order whose loop reaches back to b, in `re` order, while b is not excluded: the entries it
//     passes on the way are old acquaintances (or excluded by b a moment ago, and skipped);
// then the excluded ones go and b, if it stays, takes its place in the order of the second sort.  Conditions checked here, else the caller
// takes the general path: b shares its `re` or its (score, rb, qb) with no entry; no entry of another contig lies near b in `re` order (the
// reference's loops stop at one); at most 64 entries are within reach.  Returns false when a condition fails (nothing changed).
__device__ __forceinline__ bool incr_insert(const DevOpt &o, int &n_io, const ListRef L, const ListRef tmp, DevReg b, int *idx, int l)
{
	const int n = n_io;
	const int64_t gap = o.max_chain_gap;
	bool fail = false;
	int pos = 0, n_cand = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + l;
		bool cand = false, before = false, bad = false;
		if (i < n) {
			const int64_t xrb = L[i].rb, xre = L[i].re;
			const int xrid = L[i].rid, xsc = L[i].score, xqb = L[i].qb;
			bad = xre == b.re || (xsc == b.score && xrb == b.rb && xqb == b.qb) || (xrid != b.rid && xre > b.rb - gap - 1 && xre < b.re + gap + 65536);
			cand = xrid == b.rid && ((xre < b.re && b.rb < xre + gap) || (xre > b.re && xrb < b.re + gap));
			before = xsc > b.score || (xsc == b.score && (xrb < b.rb || (xrb == b.rb && xqb < b.qb)));
		}
		if (__ballot(bad)) fail = true;
		pos += __popcll(__ballot(before));
		const unsigned long long cm = __ballot(cand);
		if (cand) { const int slot = n_cand + __popcll(cm & ((1ull << l) - 1)); if (slot < 64) idx[slot] = i; }
		n_cand += __popcll(cm);
	}
	if (fail || n_cand > 64) return false;
	wsync();
	// lane k holds candidate k
	const bool have = l < n_cand;
	const int ci = have ? idx[l] : 0;
	int64_t xrb = 0, xre = 0; int xqb = 0, xqe = 0, xsc = 0;
	if (have) { xrb = L[ci].rb; xre = L[ci].re; xqb = L[ci].qb; xqe = L[ci].qe; xsc = L[ci].score; }
	const bool is_pred = have && xre < b.re;
	// order: predecessors by descending re (rank 0 = nearest), then successors by ascending re
	int rank = 0;
	for (int s_ = 0; s_ < n_cand; ++s_) {
		const int64_t ore = __shfl(xre, s_);
		const bool opred = ore < b.re;
		if (have && opred == is_pred && (is_pred ? ore > xre : ore < xre)) ++rank;
	}
	const int n_pred = __popcll(__ballot(is_pred));
	bool exc = false, b_exc = false;
	for (int r = 0; r < n_cand && !b_exc; ++r) {
		const bool pred_turn = r < n_pred;
		const unsigned long long who = __ballot(have && is_pred == pred_turn && rank == (pred_turn ? r : r - n_pred));
		const int k = __ffsll((long long)who) - 1;               // exactly one lane (all re differ)
		const int64_t krb = __shfl(xrb, k), kre = __shfl(xre, k);
		const int kqb = __shfl(xqb, k), kqe = __shfl(xqe, k), ksc = __shfl(xsc, k);
		// p = the later one in `re` order, q = the earlier one (bwamem.c:457-466)
		const int64_t prb = pred_turn ? b.rb : krb, pre = pred_turn ? b.re : kre, qrb = pred_turn ? krb : b.rb, qre = pred_turn ? kre : b.re;
		const int pqb = pred_turn ? b.qb : kqb, pqe = pred_turn ? b.qe : kqe, qqb = pred_turn ? kqb : b.qb, qqe = pred_turn ? kqe : b.qe;
		const int psc = pred_turn ? b.score : ksc, qsc = pred_turn ? ksc : b.score;
		const int64_t orr = qre - prb, oq = qqb < pqb ? qqe - pqb : pqe - qqb;
		const int64_t mr = qre - qrb < pre - prb ? qre - qrb : pre - prb;
		const int64_t mq = qqe - qqb < pqe - pqb ? qqe - qqb : pqe - pqb;
		if ((float)orr > o.mask_level_redun * (float)mr && (float)oq > o.mask_level_redun * (float)mq) {
			const bool p_goes = psc < qsc;
			if (pred_turn) { if (p_goes) b_exc = true; else if (l == k) exc = true; }
			else           { if (p_goes) { if (l == k) exc = true; } else b_exc = true; }
		}
	}
	const unsigned long long em = __ballot(exc);
	int m = n;
	if (em) {                                                     // drop the excluded entries, order kept
		if (exc) L[ci].qe = L[ci].qb;
		wsync();
		m = 0;
		for (int base = 0; base < n; base += 64) {
			const int i = base + l;
			const bool keep = i < n && L[i].qe > L[i].qb;
			const unsigned long long km = __ballot(keep);
			if (keep) tmp[m + __popcll(km & ((1ull << l) - 1))] = L[i];
			m += __popcll(km);
		}
		wsync();
		for (int i = l; i < m; i += 64) L[i] = tmp[i];
		wsync();
		pos = 0;
		for (int base = 0; base < m; base += 64) {
			const int i = base + l;
			bool before = false;
			if (i < m) { const int64_t yrb = L[i].rb; const int ysc = L[i].score, yqb = L[i].qb; before = ysc > b.score || (ysc == b.score && (yrb < b.rb || (yrb == b.rb && yqb < b.qb))); }
			pos += __popcll(__ballot(before));
		}
	}
	if (!b_exc) {
		for (int hi = m; hi > pos; hi -= 64) {                    // shift [pos, m) up by one, from the top, 64 at a time
			const int i = hi - 1 - l;
			DevReg v;
			const bool mv = i >= pos;
			if (mv) v = L[i];
			wsync();
			if (mv) L[i + 1] = v;
			wsync();
		}
		b.n_comp = 1;
		if (l == 0) L[pos] = b;
		++m;
		wsync();
	}
	n_io = m;
	return true;
}

// mem_matesw (bwamem_pair.c:137-206): anchor `an` (a region of one end), mate read r_m (length l_ms), mate list L (n_ma).
// P = 16: byte kernel (l_ms * a < 250), P = 8: word kernel.  Returns the new length of the mate list.
template <int P>
__device__ __forceinline__ int matesw(const PairLaunch &a, const DevReg an, int slot0, int r_m, int l_ms, const ListRef L, int n_ma, const ListRef tmp, RegKey *keys, int *idx,
                      const MsLds &m, uint8_t *slab, int l, unsigned long long &n_sw, unsigned long long &n_new, unsigned long long &n_inline, bool &clean)
{
	// clean: the list is what a mem_sort_dedup_patch without ties returned (or that plus incr_insert steps): one more call is the identity
	const DevOpt &opt = a.opt;
	const DevIndex &ix = a.ix;
	const int64_t l_pac = ix.l_pac;
	int skip = 0;
	for (int d = 0; d < 4; ++d) skip |= a.pes[d].failed ? 1 << d : 0;
	for (int base = 0; base < n_ma; base += 64) {
		const int k = base + l;
		int d = 0; bool in = false;
		if (k < n_ma) {
			int64_t dist;
			d = infer_dir(l_pac, an.rb, L[k].rb, &dist);
			in = dist >= a.pes[d].low && dist <= a.pes[d].high;
		}
		for (int dd = 0; dd < 4; ++dd) if (__ballot(in && d == dd)) skip |= 1 << dd;
	}
	if (skip == 15) return n_ma;
	const uint8_t *ms = a.seq + a.off[r_m];
	int n = 0;
	for (int r = 0; r < 4; ++r) {
		if (skip >> r & 1) continue;
		bool general = !clean;                                    // this orientation ends with the general mem_sort_dedup_patch
		const int is_rev = (r >> 1) != (r & 1);
		int64_t rb, re;
		if (ms_window(a, an, r, l_ms, rb, re)) {
			const unsigned long long tk0 = wall_clock64();
			const int tlen = (int)(re - rb);
			ssw::Res aln = { 0, -1, -1, -1, -1, -1, -1 };
			const SwRes pre = a.sw_res[slot0 + r];
			const unsigned long long tk1 = wall_clock64();
			if (pre.state == 2) {                                     // done ahead by k_matesw_sw (same anchor, orientation, window)
				aln.score = pre.score; aln.te = pre.te; aln.qe = pre.qe; aln.score2 = pre.score2; aln.te2 = pre.te2; aln.tb = pre.tb; aln.qb = pre.qb;
			} else {
				uint8_t *s_t = tlen <= m.tw_cap ? m.tw : slab;             // reference window: LDS, or this workgroup's global scratch when too wide
				__syncthreads();
				for (int i = l; i < tlen; i += 64) s_t[i] = (uint8_t)ref_base(ix, rb + i);
				if (is_rev) for (int i = l; i < l_ms; i += 64) { const uint8_t c = ms[i]; m.q[l_ms - 1 - i] = c < 4 ? 3 - c : 4; }
				else for (int i = l; i < l_ms; i += 64) m.q[i] = ms[i];
				wsync();
				const int xtra = ssw::XSUBO | ssw::XSTART | (P == 16 ? ssw::XBYTE : 0) | (opt.min_seed_len * opt.a);
				if (l < P) {                                          // one group of the wavefront runs the alignment
					const int cells = (l_ms + P - 1) / P * P;
					ssw::Work w;
					w.prof = m.prof; w.H0 = m.h; w.H1 = m.h + cells; w.E = m.h + 2 * cells; w.Hmax = m.h + 3 * cells;
					w.colmax = tlen <= m.tw_cap ? m.cm : reinterpret_cast<uint16_t*>(slab + ((size_t)tlen + 63) / 64 * 64);
					// byte kernel: at most 16 segments (249 bases); 10 covers reads up to 160 bases
					if (P == 16 && l_ms <= 160) aln = ssw::align2<P, 10>(w, l, l_ms, m.q, 1, tlen, s_t, 1, m.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra);
					else if (P == 16) aln = ssw::align2<P, 16>(w, l, l_ms, m.q, 1, tlen, s_t, 1, m.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra);
					else aln = ssw::align2<P>(w, l, l_ms, m.q, 1, tlen, s_t, 1, m.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra);
				}
				aln.score = __shfl(aln.score, 0); aln.te = __shfl(aln.te, 0); aln.qe = __shfl(aln.qe, 0); aln.score2 = __shfl(aln.score2, 0);
				aln.tb = __shfl(aln.tb, 0); aln.qb = __shfl(aln.qb, 0);
				++n_inline;
			}
			++n_sw;
			const unsigned long long tk2 = wall_clock64();
			if (l == 0) { atomicAdd(&a.counters[4], tk1 - tk0); atomicAdd(&a.counters[5], tk2 - tk1); }
			if (aln.score >= opt.min_seed_len && aln.qb >= 0) {   // something goes wrong if aln.qb < 0 (bwamem_pair.c:178)
				DevReg b;
				memset(&b, 0, sizeof b);
				b.rid = an.rid; b.is_alt = an.is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score; b.csub = aln.score2;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				const unsigned long long tki = wall_clock64();
				const bool done = !general && incr_insert(opt, n_ma, L, tmp, b, idx, l);
				if (done) { ++n_new; if (l == 0) atomicAdd(&a.counters[6], wall_clock64() - tki); }
				else {
				general = true;
				// insert b keeping the list sorted by score (bwamem_pair.c:194-199)
				int at = n_ma;
				for (int base = 0; base < n_ma; base += 64) {
					const unsigned long long lower = __ballot(base + l < n_ma && L[base + l].score < b.score);
					if (lower) { at = base + __ffsll((long long)lower) - 1; break; }
				}
				__syncthreads();
				for (int hi = n_ma; hi > at; hi -= 64) {           // shift [at, n_ma) up by one, from the top, 64 at a time
					const int i = hi - 1 - l;
					DevReg v;
					const bool mv = i >= at;
					if (mv) v = L[i];
					wsync();
					if (mv) L[i + 1] = v;
					wsync();
				}
				if (l == 0) L[at] = b;
				++n_ma; ++n_new;
				wsync();
				}
			}
			++n;
		}
		if (n && general) { const unsigned long long tk3 = wall_clock64(); bool ties; n_ma = sort_dedup_nopatch(opt, n_ma, L, tmp, keys, idx, m.stk, reinterpret_cast<unsigned*>(m.h + 128), l, a.err, ties); clean = !ties; if (l == 0) atomicAdd(&a.counters[6], wall_clock64() - tk3); }   // (m.h + 128: 256 bytes into the array -- a base register of the sort's table accesses stays inside LDS whatever offset gets folded)
	}
	return n_ma;
}

// The alignments k_pe_copy queued: one per 16-lane group, four per wavefront, byte kernel, cells in registers.  A single
// alignment is a chain of ~1 500 dependent cycles per reference base; what this kernel buys is that the (up to dozens of)
// alignments of one pair, which mem_matesw runs one after the other, proceed side by side.
// QMAX: longest mate the instantiation takes (160: ten cells per lane; 256: sixteen).  LDS decides how many alignments a CU holds: column maxima
// as bytes, the window two bases per byte and arrays sized for 160 bases make it 9.8 KB per workgroup = 4 wavefronts per SIMD (18.4 KB, 2 wavefronts with bytes, 16-bit maxima and the general sizes).
template <int QMAX>
__global__ __launch_bounds__(64) void k_matesw_sw(PairLaunch a)
{
	__shared__ int8_t s_mat[32];
	__shared__ uint8_t s_tw[4][SW_TW / 2];                       // the reference window, two bases per byte
	__shared__ uint8_t s_cm[4][SW_TW];
	__shared__ uint8_t s_q[4][QMAX];
	__shared__ int8_t s_prof[4][5 * QMAX];
	const int lane_ = lane(), g = lane_ >> 4, gl = lane_ & 15;
	if (lane_ < 25) s_mat[lane_] = a.opt.mat[lane_];
	__syncthreads();
	const int n_tasks = *a.sw_n;
	const int t = (int)blockIdx.x * 4 + g;
	if (t >= n_tasks) return;
	const int slot = a.sw_tasks[t];
	const int2 info = a.sw_info[slot];
	const int r = info.x, j = info.y >> 2, o = info.y & 3, rm = r ^ 1;
	const DevReg an = a.regs[a.reg_base[r] + j];
	const int l_ms = (int)(a.off[rm + 1] - a.off[rm]);
	const uint8_t *ms = a.seq + a.off[rm];
	int64_t rb, re;
	ms_window(a, an, o, l_ms, rb, re);                            // eligibility was established by k_pe_copy
	const int tlen = (int)(re - rb);
	const int is_rev = (o >> 1) != (o & 1);
	for (int i = gl; 2 * i < tlen; i += 16)
		s_tw[g][i] = (uint8_t)(ref_base(a.ix, rb + 2 * i) | (2 * i + 1 < tlen ? ref_base(a.ix, rb + 2 * i + 1) : 0) << 4);
	if (is_rev) for (int i = gl; i < l_ms; i += 16) { const uint8_t c = ms[i]; s_q[g][l_ms - 1 - i] = c < 4 ? 3 - c : 4; }
	else for (int i = gl; i < l_ms; i += 16) s_q[g][i] = ms[i];
	__threadfence_block();
	ssw::Work w;
	w.prof = s_prof[g]; w.H0 = w.H1 = w.E = w.Hmax = nullptr; w.colmax = nullptr; w.colmax8 = s_cm[g];
	const int xtra = ssw::XSUBO | ssw::XSTART | ssw::XBYTE | (a.opt.min_seed_len * a.opt.a);
	ssw::Res aln;
	if (QMAX <= 160 || l_ms <= 160) aln = ssw::align2<16, 10>(w, lane_, l_ms, s_q[g], 1, tlen, s_tw[g], 1, s_mat, a.opt.o_del, a.opt.e_del, a.opt.o_ins, a.opt.e_ins, xtra, true);
	else aln = ssw::align2<16, 16>(w, lane_, l_ms, s_q[g], 1, tlen, s_tw[g], 1, s_mat, a.opt.o_del, a.opt.e_del, a.opt.o_ins, a.opt.e_ins, xtra, true);
	if (gl == 0) { SwRes o_ = { 2, aln.score, aln.te, aln.qe, aln.score2, aln.te2, aln.tb, aln.qb }; a.sw_res[slot] = o_; }
}

// The same for mates of 250 bases x a and more (up to SW8_QMAX bases): the word kernel (ksw_i16: ksw_align2 takes it when KSW_XBYTE is not set,
// bwamem_pair.c:171), eight lanes per alignment, eight alignments per wavefront, 32 cells per lane in registers.
__global__ __launch_bounds__(64) void k_matesw_sw8(PairLaunch a)
{
	__shared__ int8_t s_mat[32];
	__shared__ uint8_t s_tw[8][SW_TW / 2];
	__shared__ uint16_t s_cm[8][SW_TW];
	__shared__ uint8_t s_q[8][SW8_QMAX];
	__shared__ int8_t s_prof[8][5 * SW8_QMAX];
	const int lane_ = lane(), g = lane_ >> 3, gl = lane_ & 7;
	if (lane_ < 25) s_mat[lane_] = a.opt.mat[lane_];
	__syncthreads();
	const int n_tasks = *a.sw_n8;
	const int t = (int)blockIdx.x * 8 + g;
	if (t >= n_tasks) return;
	const int slot = a.sw_tasks[a.sw_cap - 1 - t];
	const int2 info = a.sw_info[slot];
	const int r = info.x, j = info.y >> 2, o = info.y & 3, rm = r ^ 1;
	const DevReg an = a.regs[a.reg_base[r] + j];
	const int l_ms = (int)(a.off[rm + 1] - a.off[rm]);
	const uint8_t *ms = a.seq + a.off[rm];
	int64_t rb, re;
	ms_window(a, an, o, l_ms, rb, re);                            // eligibility was established by k_pe_copy
	const int tlen = (int)(re - rb);
	const int is_rev = (o >> 1) != (o & 1);
	for (int i = gl; 2 * i < tlen; i += 8)
		s_tw[g][i] = (uint8_t)(ref_base(a.ix, rb + 2 * i) | (2 * i + 1 < tlen ? ref_base(a.ix, rb + 2 * i + 1) : 0) << 4);
	if (is_rev) for (int i = gl; i < l_ms; i += 8) { const uint8_t c = ms[i]; s_q[g][l_ms - 1 - i] = c < 4 ? 3 - c : 4; }
	else for (int i = gl; i < l_ms; i += 8) s_q[g][i] = ms[i];
	__threadfence_block();
	ssw::Work w;
	w.prof = s_prof[g]; w.H0 = w.H1 = w.E = w.Hmax = nullptr; w.colmax = s_cm[g]; w.colmax8 = nullptr;
	const int xtra = ssw::XSUBO | ssw::XSTART | (a.opt.min_seed_len * a.opt.a);
	const ssw::Res aln = ssw::align2<8, SW8_QMAX / 8>(w, lane_, l_ms, s_q[g], 1, tlen, s_tw[g], 1, s_mat, a.opt.o_del, a.opt.e_del, a.opt.o_ins, a.opt.e_ins, xtra, true);
	if (gl == 0) { SwRes o_ = { 2, aln.score, aln.te, aln.qe, aln.score2, aln.te2, aln.tb, aln.qb }; a.sw_res[slot] = o_; }
}

// BIG: the pairs with a list beyond MS_LIST regions (their lists stay in global memory): this instantiation has no list arrays in LDS, so
// eight of its workgroups fit a CU instead of two -- their work is chains of dependent accesses, and a human-like repeat load is all BIG pairs
template <int P, bool BIG>
__global__ __launch_bounds__(64, BIG ? 2 : 1) void k_matesw(PairLaunch a)
{
	constexpr int LDS_LIST = BIG ? 1 : MS_LIST;
	constexpr int CELLS = P == 16 ? 256 : (MS_MAXQ + 7) / 8 * 8;
	__shared__ uint8_t s_q[MS_MAXQ + 8];
	__shared__ int8_t s_mat[32];
	__shared__ int8_t s_prof[5 * CELLS];
	__shared__ int16_t s_h[4 * CELLS];
	__shared__ int s_stk[3 * 80];
	__shared__ uint8_t s_tw[4096];
	__shared__ uint16_t s_cm[4096];
	// 16-byte aligned: the lists are reached through generic pointers (LDS or global slots), i.e. flat 16-byte accesses
	__shared__ GuardedLds<LDS_LIST> s_list_g, s_tmp_g;         // (a spare record in front and behind: see ListRef)
	__shared__ __attribute__((aligned(16))) RegKey s_keys[LDS_LIST];
	const ListRef s_list = ListRef::lds(s_list_g), s_tmp = ListRef::lds(s_tmp_g);
	__shared__ int s_idx[2 * LDS_LIST];
	const int l = lane();
	if (l < 25) s_mat[l] = a.opt.mat[l];
	__syncthreads();
	uint8_t *slab = a.slab + (size_t)blockIdx.x * a.slab_stride;
	unsigned long long n_sw = 0, n_new = 0, max_sw = 0, n_inline = 0;
	const int n_resc = *a.resc_n;
	// the pairs come from a queue, the ones with the longest lists first (k_resc_order): a pair inside a repeat family costs a thousand times an
	// ordinary one, and a fixed share per workgroup left most of the kernel's duration to the unluckiest workgroup
	for (;;) {
		int it = 0;
		if (l == 0) it = (int)atomicAdd(a.queue + (P == 16 ? 0 : 1) + (BIG ? 2 : 0), 1u);
		it = __shfl(it, 0);
		if (it >= n_resc) break;
		const int p = a.resc_list[it];
		if ((a.pe_cap[p << 1] > MS_LIST || a.pe_cap[p << 1 | 1] > MS_LIST) != BIG) continue;   // the other instantiation's pair
		const unsigned long long sw_before = n_sw, tp0 = wall_clock64();
		int n_list[2] = { a.pe_n[p << 1], a.pe_n[p << 1 | 1] };
		// sort scratch of a list lives behind its slots' spare copy: tmp list, keys and index arrays sized by the capacity
		for (int i = 0; i < 2; ++i) {
			const int r = p << 1 | i, rm = r ^ 1;
			const int l_ms = (int)(a.off[rm + 1] - a.off[rm]);
			if ((P == 16) != (l_ms * a.opt.a < 250)) continue;    // the other instantiation takes this mate length
			if (a.nb[r] == 0) continue;
			// the mate's list is worked on in LDS when its capacity fits (the insert / sort / dedup steps are chains of dependent
			// accesses: ~1 ms per rescue through global memory, measured), else in its global slots
			DevReg *G = a.pe_regs + a.pe_base[rm];
			const bool in_lds = !BIG;
			const DevReg *snap = a.regs + a.reg_base[r];            // b[i]: the end's own regions as mem_align1_core left them
			// two call sites on purpose: each is compiled for its own address space (LDS / global).  Through one generic pointer the
			// list code became flat instructions with folded offsets (base = &L[i-1], offset +80); the hardware picks the aperture
			// from the base alone, and &s_list[-1] lies outside the LDS aperture: a memory violation (seen under rocgdb).
			if (in_lds) {
				__syncthreads();
				for (int k = l; k < n_list[i ^ 1]; k += 64) s_list[k] = G[k];
				wsync();
				const MsLds m = { s_q, s_mat, s_prof, s_h, s_keys, s_idx, s_stk, s_tw, 4096, s_cm };
				bool clean = false;
				for (int j = 0; j < a.nb[r]; ++j)
					n_list[i ^ 1] = matesw<P>(a, snap[j], (int)a.sw_base[r] + 4 * j, rm, l_ms, s_list, n_list[i ^ 1], s_tmp, s_keys, s_idx, m, slab, l, n_sw, n_new, n_inline, clean);
			} else {
				DevReg *tmp = a.pe_tmp + a.pe_base[rm];
				RegKey *keys = reinterpret_cast<RegKey*>(a.pe_keys) + a.pe_base[rm];
				int *idx = a.pe_idx + 2 * a.pe_base[rm];
				const MsLds m = { s_q, s_mat, s_prof, s_h, keys, idx, s_stk, s_tw, 4096, s_cm };
				bool clean = false;
				for (int j = 0; j < a.nb[r]; ++j)
					n_list[i ^ 1] = matesw<P>(a, snap[j], (int)a.sw_base[r] + 4 * j, rm, l_ms, ListRef::global(G), n_list[i ^ 1], ListRef::global(tmp), keys, idx, m, slab, l, n_sw, n_new, n_inline, clean);
			}
			if (in_lds) { wsync(); for (int k = l; k < n_list[i ^ 1]; k += 64) G[k] = s_list[k]; wsync(); }
		}
		if (l == 0) {
			const int lm0 = (int)(a.off[(p << 1) + 1] - a.off[p << 1]), lm1 = (int)(a.off[(p << 1) + 2] - a.off[(p << 1) + 1]);
			// each instantiation updates the list(s) it rescued into: list of read 1 is rescued with read 1's sequence (i = 0), list 0 with read 0's
			if ((P == 16) == (lm1 * a.opt.a < 250)) a.pe_n[p << 1 | 1] = n_list[1];
			if ((P == 16) == (lm0 * a.opt.a < 250)) a.pe_n[p << 1] = n_list[0];
		}
		max_sw = max_sw > n_sw - sw_before ? max_sw : n_sw - sw_before;
		if (l == 0) atomicMax(&a.counters[7], wall_clock64() - tp0);
		__syncthreads();
	}
	if (l == 0) { if (n_sw) { atomicAdd(&a.counters[0], n_sw); atomicAdd(&a.counters[1], n_new); atomicMax(&a.counters[2], max_sw); atomicAdd(&a.counters[8], n_inline); } if (blockIdx.x == 0) atomicMax(&a.counters[3], (unsigned long long)n_resc); }
}

// ---------------------------------------------------------------------------------------------------- mem_pair + mem_sam_pe decisions
struct PKey { uint64_t x, y; };
__device__ __forceinline__ bool pk_lt(const PKey &a, const PKey &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }
__device__ __forceinline__ PKey wmax_pk(PKey v)
{
	for (int d = 32; d; d >>= 1) {
		PKey o;
		o.x = (uint64_t)__shfl_xor((long long)v.x, d); o.y = (uint64_t)__shfl_xor((long long)v.y, d);
		if (pk_lt(v, o)) v = o;
	}
	return v;
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))

// One pair per wavefront.
__global__ __launch_bounds__(64) void k_pair(PairLaunch a)
{
	const int l = lane();
	const int p = a.subset == 2 ? a.resc_list[blockIdx.x] : (int)blockIdx.x;
	if (a.subset == 1 && a.resc_flag[p]) return;                 // finalised after its rescue, by the second launch
	const DevOpt &opt = a.opt;
	const int64_t l_pac = a.ix.l_pac;
	const int r0 = p << 1;
	FinReg *f[2] = { a.fregs_w + a.pe_base[r0], a.fregs_w + a.pe_base[r0 | 1] };
	const int nn[2] = { a.freg_n[r0], a.freg_n[r0 | 1] }, n_pri[2] = { a.n_pri[r0], a.n_pri[r0 | 1] };
	uint8_t *need[2] = { a.need + a.pe_base[r0], a.need + a.pe_base[r0 | 1] };
	int *owner[2] = { a.xa_owner + a.pe_base[r0], a.xa_owner + a.pe_base[r0 | 1] };
	int *scr[2] = { a.scr + 4 * a.pe_base[r0], a.scr + 4 * a.pe_base[r0 | 1] };
	PeRead pr[2];
	for (int i = 0; i < 2; ++i) { pr[i].mode = 0; pr[i].h_reg = -1; pr[i].alt_reg = -1; pr[i].mapq = 0; pr[i].extra_flag = 1; pr[i].pad[0] = pr[i].pad[1] = pr[i].pad[2] = 0; }
	bool paired = false;
	int z[2] = { 0, 0 }, q_se[2] = { 0, 0 }, extra_flag = 1;

	if (!(opt.flag & BWAHIP_F_NOPAIRING) && n_pri[0] && n_pri[1]) {
		// ---- mem_pair (bwamem_pair.c:208-272).  v: the primary-assembly regions of both ends, sorted by (contig, forward position)
		const int nv = n_pri[0] + n_pri[1];
		PKey *v = reinterpret_cast<PKey*>(scr[0]);                  // 16 B x (cap0 + cap1) >= nv entries (the two reads' scratch is contiguous)
		auto make_key = [&](int t) {
			const int r = t < n_pri[0] ? 0 : 1, i = r ? t - n_pri[0] : t;
			const FinReg &e = (r ? f[1] : f[0])[i];                  // (no dynamic index into the local pointer array: that would put it into scratch memory)
			PKey k;
			k.x = e.rb < l_pac ? (uint64_t)e.rb : (uint64_t)((l_pac << 1) - 1 - e.rb);
			k.x = (uint64_t)e.rid << 32 | (k.x - (uint64_t)a.ix.anns[e.rid].offset);
			k.y = (uint64_t)e.score << 32 | (uint64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
			return k;
		};
		if (nv <= 64) {                                             // one key per lane; the others come by shuffle (rank sort: the keys are unique)
			PKey kt = { ~0ull, ~0ull };
			if (l < nv) kt = make_key(l);
			int rank = 0;
			for (int u = 0; u < nv; ++u) {
				PKey ku;
				ku.x = (uint64_t)__shfl((long long)kt.x, u); ku.y = (uint64_t)__shfl((long long)kt.y, u);
				rank += pk_lt(ku, kt) ? 1 : 0;
			}
			if (l < nv) v[rank] = kt;
		} else {
			// a pair inside a repeat family (hundreds of regions per end; its sort would set the duration of the whole kernel): the keys are
			// written once and ordered by a bitonic network in the second region array, which nothing uses any more
			int m = 128;
			while (m < nv) m <<= 1;
			PKey *w = reinterpret_cast<PKey*>(a.fregs_tmp + a.pe_base[r0]);      // 96 B x (regions of both ends) >= 16 B x 2 nv >= 16 B x m
			for (int t = l; t < m; t += 64) w[t] = t < nv ? make_key(t) : PKey{ ~0ull, ~0ull };
			wsync();
			for (int k = 2; k <= m; k <<= 1) {
				for (int j = k >> 1, lj = 31 - __clz(k >> 1); j > 0; j >>= 1, --lj) {
					for (int t = l; t < (m >> 1); t += 64) {
						const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), q = i + j;
						const PKey x = w[i], y = w[q];
						if (((i & k) == 0) ? pk_lt(y, x) : pk_lt(x, y)) { w[i] = y; w[q] = x; }
					}
					wsync();
				}
			}
			for (int t = l; t < nv; t += 64) v[t] = w[t];
		}
		wsync();
		const int id = (int)((a.n_processed >> 1) + p);
		// pass 1: the two largest elements of u (every admissible pair (k < i) with its score q and tie-breaking hash), and their number
		uint64_t bx = 0, by = 0, sx = 0, sy = 0;                    // best / second, field by field
		int cnt = 0;
		// every admissible pair (k < i) of the sorted list (bwamem_pair.c:232-246) as (ux, uy) = (score << 32 | hash, k << 32 | i).  A macro, not a
		// lambda: what a lambda's callers capture by reference ends up in scratch memory unless everything is inlined
#define FOR_PAIRS(BODY) \
		for (int i = l; i < nv; i += 64) { \
			const PKey vi = v[i]; \
			for (int r = 0; r < 2; ++r) { \
				const int dir = r << 1 | (int)(vi.y >> 1 & 1); \
				if (a.pes[dir].failed) continue; \
				const int which = r << 1 | (int)((vi.y & 1) ^ 1); \
				for (int k = i - 1; k >= 0; --k) { \
					const PKey vk = v[k]; \
					if ((int)(vk.y & 3) != which) continue; \
					const int64_t dist = (int64_t)vi.x - (int64_t)vk.x; \
					if (dist > a.pes[dir].high) break; \
					if (dist < a.pes[dir].low) continue; \
					int q = (int)((double)((vi.y >> 32) + (vk.y >> 32)) + a.pair_tab[a.tab_off[dir] + (int)(dist - a.pes[dir].low)] * opt.a + .499); \
					if (q < 0) q = 0; \
					const uint64_t uy = (uint64_t)k << 32 | (uint64_t)i; \
					const uint64_t ux = (uint64_t)q << 32 | (hash_64(uy ^ (uint64_t)(int64_t)(id << 8)) & 0xffffffffU); \
					BODY \
				} \
			} \
		}
		FOR_PAIRS({
			++cnt;
			// value selects only: "if (c) a = u; else b = u;" is turned into a store through a selected ADDRESS, which pins a and b in scratch
			const bool top1 = cnt == 1 || bx < ux || (bx == ux && by < uy);
			const bool top2 = !top1 && (cnt == 2 || sx < ux || (sx == ux && sy < uy));
			const uint64_t nsx = top1 ? (cnt > 1 ? bx : sx) : top2 ? ux : sx; const uint64_t nsy = top1 ? (cnt > 1 ? by : sy) : top2 ? uy : sy;
			bx = top1 ? ux : bx; by = top1 ? uy : by; sx = nsx; sy = nsy;
		})

		// (best, second) per lane -> of the wavefront.  A lane with no pair offers {0,0}; real elements have y >= 1 (i >= 1)
		const int n_u = wsum(cnt);
		if (n_u > 0) {
			const int my_cnt = cnt;
			// (selected field by field: a conditional between whole structs is compiled as a choice between their addresses, which
			// pins them in scratch memory)
			PKey mine;
			mine.x = my_cnt ? bx : 0; mine.y = my_cnt ? by : 0;
			const PKey top = wmax_pk(mine);
			const bool holder = my_cnt && mine.x == top.x && mine.y == top.y;
			PKey off2;
			off2.x = holder ? (my_cnt > 1 ? sx : 0) : mine.x; off2.y = holder ? (my_cnt > 1 ? sy : 0) : mine.y;
			const PKey sec = wmax_pk(off2);
			int tmp = opt.a + opt.b;
			tmp = tmp > opt.o_del + opt.e_del ? tmp : opt.o_del + opt.e_del;
			tmp = tmp > opt.o_ins + opt.e_ins ? tmp : opt.o_ins + opt.e_ins;
			const int ti = (int)(top.y & 0xffffffffu), tk = (int)(top.y >> 32);
			{ const PKey a_ = v[ti], b_ = v[tk]; const int za = (int)((a_.y & 0xffffffffu) >> 2), zb = (int)((b_.y & 0xffffffffu) >> 2);
			  if (a_.y & 1) z[1] = za; else z[0] = za;
			  if (b_.y & 1) z[1] = zb; else z[0] = zb; }
			const int o = (int)(top.x >> 32);
			int subo = n_u > 1 ? (int)(sec.x >> 32) : 0, n_sub = 0;
			{   // pass 2: bwamem_pair.c:266-268 -- all but the top element whose score is within tmp of subo
				int c2 = 0;
				FOR_PAIRS({ (void)uy; if (subo - (int)(ux >> 32) <= tmp) ++c2; })
#undef FOR_PAIRS
				n_sub = wsum(c2) - 1;
			}
			if (o > 0) {
				// ---- bwamem_pair.c:312-365
				int is_multi[2];
				for (int i = 0; i < 2; ++i) {                         // bwamem_pair.c:314-317: any further primary hit above T?  64 regions per step
					is_multi[i] = 0;
					for (int base = 1; base < n_pri[i] && !is_multi[i]; base += 64) {
						const int j = base + l;
						const bool hit = j < n_pri[i] && f[i][j].secondary < 0 && f[i][j].score >= opt.T;
						if (__ballot(hit)) is_multi[i] = 1;
					}
				}
				if (!(is_multi[0] || is_multi[1])) {
					paired = true;
					const int score_un = f[0][0].score + f[1][0].score - opt.pen_unpaired;
					subo = subo > score_un ? subo : score_un;
					int q_pe = RAW_MAPQ(o - subo, opt.a);
					if (n_sub > 0) q_pe -= (int)(4.343 * a.logtab[n_sub + 1 < BWAHIP_LOGTAB_N ? n_sub + 1 : BWAHIP_LOGTAB_N - 1] + .499);
					if (n_sub + 1 >= BWAHIP_LOGTAB_N && l == 0) atomicExch(a.err, 61);
					if (q_pe < 0) q_pe = 0;
					if (q_pe > 60) q_pe = 60;
					q_pe = (int)(q_pe * (1. - .5 * (f[0][0].frac_rep + f[1][0].frac_rep)) + .499);
					int bad = 0;
					if (o > score_un) {                              // paired alignment is preferred
						for (int i = 0; i < 2; ++i) {
							FinReg c = f[i][z[i]];
							if (c.secondary >= 0) {
								c.sub = f[i][c.secondary].score; c.secondary = -2;
								if (l == 0) { f[i][z[i]].sub = c.sub; f[i][z[i]].secondary = -2; }
							}
							q_se[i] = approx_mapq_se(opt, a.logtab, c, &bad);
						}
						wsync();
						q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
						q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
						extra_flag |= 2;
						for (int i = 0; i < 2; ++i) {
							const int cap = RAW_MAPQ(f[i][z[i]].score - f[i][z[i]].csub, opt.a);
							q_se[i] = q_se[i] < cap ? q_se[i] : cap;
						}
					} else {                                          // the unpaired alignment is preferred
						z[0] = z[1] = 0;
						q_se[0] = approx_mapq_se(opt, a.logtab, f[0][0], &bad);
						q_se[1] = approx_mapq_se(opt, a.logtab, f[1][0], &bad);
					}
					if (bad && l == 0) atomicExch(a.err, 62);
					for (int i = 0; i < 2; ++i) {                     // bwamem_pair.c:352-360
						const int k = f[i][z[i]].secondary_all;
						if (k >= 0 && k < n_pri[i]) {
							wsync();
							for (int j = l; j < nn[i]; j += 64) if (f[i][j].secondary_all == k || j == k) f[i][j].secondary_all = z[i];
							wsync();
							if (l == 0) f[i][z[i]].secondary_all = -1;
							wsync();
						}
					}
				}
			}
		}
	}

	if (paired) {
		for (int i = 0; i < 2; ++i) {
			const int n = nn[i];
			// XA membership exactly as mem_gen_alt computes it over the whole list; only the members of the printed records are kept
			int n_task, n_rec;
			select_records(opt, n, f[i], need[i], owner[i], scr[i], l, n_task, n_rec);
			wsync();
			int alt_reg = -1;
			if (n_pri[i] < n) {
				const FinReg &q = f[i][n_pri[i]];
				if (!(q.score < opt.T || q.secondary >= 0 || !q.is_alt)) alt_reg = n_pri[i];
			}
			int cnt_task = 0;
			for (int base = 0; base < n; base += 64) {
				const int j = base + l;
				int nd = 0;
				if (j < n) {
					if (j == z[i] || j == alt_reg) nd |= NEED_REC;
					if ((need[i][j] & NEED_XA) && (owner[i][j] == z[i] || owner[i][j] == alt_reg)) nd |= NEED_XA; else owner[i][j] = -1;
					need[i][j] = (uint8_t)nd;
				}
				cnt_task += __popcll(__ballot(nd != 0));
			}
			pr[i].mode = 1; pr[i].h_reg = z[i]; pr[i].alt_reg = alt_reg; pr[i].mapq = q_se[i]; pr[i].extra_flag = extra_flag;
			if (l == 0) { a.task_n[r0 | i] = cnt_task; a.rec_n[r0 | i] = 1 + (alt_reg >= 0 ? 1 : 0); }
		}
	} else {
		// ---- no pairing (bwamem_pair.c:397-418): each end like a single-end read, with the mate's best hit attached
		for (int i = 0; i < 2; ++i) {
			const int n = nn[i];
			int which = -1;
			if (n > 0) {
				if (f[i][0].score >= opt.T) which = 0;
				else if (n_pri[i] < n && f[i][n_pri[i]].score >= opt.T) which = n_pri[i];
			}
			int n_task = 0, n_rec = 0;
			if (n > 0) {
				select_records(opt, n, f[i], need[i], owner[i], scr[i], l, n_task, n_rec);
				wsync();
				if (which >= 0 && need[i][which] == 0) { ++n_task; if (l == 0) need[i][which] = NEED_H; }   // h[i] of a region that prints nothing itself
				wsync();
			}
			pr[i].mode = 0; pr[i].h_reg = which; pr[i].extra_flag = 1;
			if (l == 0) { a.task_n[r0 | i] = n_task; a.rec_n[r0 | i] = n_rec; }
		}
	}
	if (l == 0) { a.pe_read[r0] = pr[0]; a.pe_read[r0 | 1] = pr[1]; }
}

} // namespace

size_t matesw_slab_bytes(int64_t window)                            // window: the widest reference window (high - low + mate length)
{
	const size_t win = (size_t)window + 256;
	return ((win + 63) / 64 * 64) + win * 2 + 256;                      // + one 16-bit column maximum per window base
}

// the rescue list ordered by the length of the pair's lists (the cost of mem_matesw's list work grows with it), longest first
__global__ void k_resc_cost(PairLaunch a, int n_resc, int *cost)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_resc) return;
	const int p = a.resc_list[i];
	cost[i] = a.pe_n[p << 1] + a.pe_n[p << 1 | 1];
}
__global__ void k_resc_apply(int n_resc, const int *list, const int *perm, int *out)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_resc) out[i] = list[perm[i]];
}
// scratch: 3 * n_resc + 8 ints; on return a.resc_list[0..n_resc) is in the new order
int launch_resc_order(const PairLaunch &a, int n_resc, int *scratch, hipStream_t st)
{
	if (n_resc <= 1) return 0;
	int *cost = scratch + 8, *perm = cost + n_resc, *tmp = perm + n_resc;
	hipLaunchKernelGGL(k_resc_cost, dim3((n_resc + 255) / 256), dim3(256), 0, st, a, n_resc, cost);
	const int rc = launch_order(n_resc, cost, 1024, 256, 64, perm, scratch, st);
	if (rc) return rc;
	hipLaunchKernelGGL(k_resc_apply, dim3((n_resc + 255) / 256), dim3(256), 0, st, n_resc, a.resc_list, perm, tmp);
	if (hipMemcpyAsync(a.resc_list, tmp, (size_t)n_resc * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return BWAHIP_ENODEV;
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_pestat(const PairLaunch &a, hipStream_t st)
{
	const int np = a.n_reads >> 1;
	if (np <= 0) return 0;
	hipLaunchKernelGGL(k_pestat, dim3((np + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_pe_prepare(const PairLaunch &a, hipStream_t st)
{
	const int np = a.n_reads >> 1;
	if (np <= 0) return 0;
	hipLaunchKernelGGL(k_pe_prepare, dim3((np + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_pe_copy(const PairLaunch &a, hipStream_t st)
{
	const int np = a.n_reads >> 1;
	if (np <= 0) return 0;
	hipLaunchKernelGGL(k_pe_copy, dim3((np + 255) / 256), dim3(256), 0, st, a);
	hipLaunchKernelGGL(k_pe_copy_big, dim3((np + 3) / 4), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_matesw(const PairLaunch &a, int grid, hipStream_t st)
{
	if (grid <= 0) return 0;
	hipLaunchKernelGGL((k_matesw<16, true>), dim3(grid), dim3(64), 0, st, a);    // mates under 250 bases: byte kernel; the pairs with long lists first
	hipLaunchKernelGGL((k_matesw<16, false>), dim3(grid), dim3(64), 0, st, a);
	hipLaunchKernelGGL((k_matesw<8, true>), dim3(grid), dim3(64), 0, st, a);     // longer mates: word kernel (after the byte one: a pair may need both)
	hipLaunchKernelGGL((k_matesw<8, false>), dim3(grid), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_matesw_sw(const PairLaunch &a, int n_tasks, int n_tasks8, int max_len, hipStream_t st)
{
	if (n_tasks8 > 0) hipLaunchKernelGGL(k_matesw_sw8, dim3((n_tasks8 + 7) / 8), dim3(64), 0, st, a);
	if (n_tasks <= 0) return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
	if (max_len <= 160) hipLaunchKernelGGL(k_matesw_sw<160>, dim3((n_tasks + 3) / 4), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_matesw_sw<256>, dim3((n_tasks + 3) / 4), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_pair(const PairLaunch &a, int n_listed, hipStream_t st)
{
	const int np = a.subset == 2 ? n_listed : a.n_reads >> 1;
	if (np <= 0) return 0;
	hipLaunchKernelGGL(k_pair, dim3(np), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
