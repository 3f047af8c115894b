// K6..K8 -- finalisation of single-end reads on the GPU (the reference's worker2, bwamem.c:1197-1213):
//   k_mark   mem_mark_primary_se (bwamem.c:500-565), mem_reorder_primary5 (bwamem.c:988), and the selection mem_reg2sam
//            (bwamem.c:1013-1059) and mem_gen_alt (bwamem_extra.c:124-169) make: which regions print a record, which are
//            listed in an XA tag.  One read per wavefront; the pairwise overlap tests run one kept region per lane.
//   k_cigar  mem_reg2aln (bwamem.c:1099-1170) per selected region: mem_approx_mapq_se (bwamem.c:962), infer_bw (799),
//            bwa_gen_cigar2 (bwa.c:261-347) with ksw_global2 + backtrack (ksw.c:504-606) on the wavefront (same
//            row-per-step DP as k_extend, direction bits kept in LDS), NM / MD, leading/trailing deletion squeeze, clips.
//   (k_sam.hip holds K9, the SAM text.)
// Sorting: both sorts of mem_mark_primary_se order by keys that contain hash_64(id+i), a bijection of the region index,
// so no two keys are equal and ANY correct sort returns the reference's (unstable) introsort permutation: rank sort.
// Floating point: mapQ uses log(l), log(sub_n+1), log(seedcov) of integers -- read from a table the host filled with
// glibc's log -- and plain IEEE double arithmetic otherwise (-ffp-contract=off).  Integer DP: MFMA not applicable.
#include "bwahip_internal.h"
#include "wave_dev.h"
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

struct SortKey { int score, is_alt; uint64_t hash; };
// mem_ars_hash (bwamem.c:404): score desc, is_alt asc, hash asc;  mem_ars_hash2 (bwamem.c:406): is_alt asc, score desc, hash asc
template <int MODE> __device__ __forceinline__ bool key_lt(const SortKey &a, const SortKey &b)
{
	if (MODE == 0) return a.score > b.score || (a.score == b.score && (a.is_alt < b.is_alt || (a.is_alt == b.is_alt && a.hash < b.hash)));
	return a.is_alt < b.is_alt || (a.is_alt == b.is_alt && (a.score > b.score || (a.score == b.score && a.hash < b.hash)));
}

// mem_mark_primary_se_core (bwamem.c:500-526) on f[0..n).  The reference takes the regions in order and tests each against the kept
// (non-secondary) ones before it, stopping at the FIRST it overlaps significantly.  Turned around here: the next kept region j is the
// first one not marked yet; every later unmarked region is tested against j at once, one per lane.  A region still ends up as the
// secondary of the first kept region it overlaps, kept regions are the same set, and j's sub (the score of the first region it
// absorbed) and sub_n (how many of them count, bwamem.c:517-518) do not depend on the order -- but the number of dependent steps is
// the number of KEPT regions (a handful) instead of the number of regions (hundreds for a read inside a repeat family).
__device__ void mark_core(const DevOpt &opt, int n, FinReg *f, int *z, int l)
{
	int tmp = opt.a + opt.b;
	tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
	tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
	if (n <= 0) return;
	(void)z;
	int j = 0;                                                  // region 0 is kept (bwamem.c:506)
	while (j >= 0) {
		const int jqb = f[j].qb, jqe = f[j].qe, jsc = f[j].score, jalt = f[j].is_alt;
		int first_i = 0x7fffffff, n_cnt = 0, next = 0x7fffffff;
		for (int base = j + 1; base < n; base += 64) {
			const int i = base + l;
			bool sig = false, counts = false, open = false;
			if (i < n && f[i].secondary < 0) {
				const int iqb = f[i].qb, iqe = f[i].qe;
				const int b_max = jqb > iqb ? jqb : iqb, e_min = jqe < iqe ? jqe : iqe;
				if (e_min > b_max) {
					const int min_l = iqe - iqb < jqe - jqb ? iqe - iqb : jqe - jqb;
					if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) sig = true;
				}
				if (sig) { f[i].secondary = j; counts = jsc - f[i].score <= tmp && (jalt || !f[i].is_alt); }
				else open = true;
			}
			const unsigned long long ms = __ballot(sig), mo = __ballot(open);
			if (ms && first_i == 0x7fffffff) first_i = base + __ffsll((long long)ms) - 1;
			n_cnt += __popcll(__ballot(counts));
			if (mo && next == 0x7fffffff) next = base + __ffsll((long long)mo) - 1;
		}
		wsync();
		if (first_i != 0x7fffffff && l == 0) {
			if (f[j].sub == 0) f[j].sub = f[first_i].score;
			f[j].sub_n += n_cnt;
		}
		wsync();
		j = next == 0x7fffffff ? -1 : next;                       // the first region left unmarked is kept too (nothing kept before it overlaps it)
	}
}

// Sort f[0..n) into g[0..n) by MODE's order; the keys (unique) go to `keys`.  Up to 128 regions: rank sort, every lane counts the keys
// below its own (n^2 / 64 comparisons per lane).  Longer lists -- a read inside a repeat family has hundreds of regions, and its sort
// would set the duration of the whole kernel -- take a bitonic network over an index array (kept in g's memory, which is only written
// at the very end): log^2 steps of n / 128 compare-exchanges per lane.
template <int MODE> __device__ void rank_sort(int n, const FinReg *f, FinReg *g, SortKey *keys, int l)
{
	for (int i = l; i < n; i += 64) { keys[i].score = f[i].score; keys[i].is_alt = f[i].is_alt; keys[i].hash = f[i].hash; }
	wsync();
	if (n <= 128) {
		for (int i = l; i < n; i += 64) {
			const SortKey ki = keys[i];
			int rank = 0;
			for (int j = 0; j < n; ++j) rank += key_lt<MODE>(keys[j], ki) ? 1 : 0;
			g[rank] = f[i];
		}
		wsync();
		return;
	}
	int m = 256;
	while (m < n) m <<= 1;
	int *perm = reinterpret_cast<int*>(g);                      // m <= 2n ints fit in n regions of 96 bytes
	for (int p = l; p < m; p += 64) perm[p] = p < n ? p : -1;    // -1: padding, sorts behind everything
	wsync();
	auto before = [&](int x, int y) { return x >= 0 && (y < 0 || key_lt<MODE>(keys[x], keys[y])); };
	for (int k = 2; k <= m; k <<= 1) {
		for (int j = k >> 1, lj = 31 - __clz(k >> 1); j > 0; j >>= 1, --lj) {
			for (int t = l; t < (m >> 1); t += 64) {
				const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), q = i + j;
				const int x = perm[i], y = perm[q];
				const bool up = (i & k) == 0;
				if (up ? before(y, x) : before(x, y)) { perm[i] = y; perm[q] = x; }
			}
			wsync();
		}
	}
	// the order into the (now free) key array, then the regions to their places -- g is overwritten from here on
	int *order = reinterpret_cast<int*>(keys);
	for (int p = l; p < n; p += 64) order[p] = perm[p];
	wsync();
	for (int p = l; p < n; p += 64) g[p] = f[order[p]];
	wsync();
}

// One read per wavefront.  PLAN: also select records / XA members (single-end output path).
template <bool PLAN>
__global__ __launch_bounds__(64) void k_mark(FinLaunch a)
{
	const int l = lane();
	const int r = a.subset == 2 ? (a.resc_pairs[blockIdx.x >> 1] << 1 | (int)(blockIdx.x & 1)) : (int)blockIdx.x;
	if (a.subset == 1 && a.resc_flag[r >> 1]) return;           // finalised after its rescue, by the second launch
	const DevOpt &opt = a.opt;
	const int n = a.reg_n[r];
	const int64_t rb0 = a.reg_base[r];
	// Reads with up to MK_LDS regions (all but a few) are marked in LDS and written to HBM once at the end: the work is a chain of small
	// dependent steps (sort, pairwise overlap tests, index fix-ups), each a memory round trip -- in HBM that chain is what the kernel waits for
	constexpr int MK_LDS = 32;
	__shared__ __attribute__((aligned(16))) FinReg s_f[MK_LDS], s_g[MK_LDS];
	__shared__ __attribute__((aligned(16))) SortKey s_keys[MK_LDS];
	__shared__ int s_z[2 * MK_LDS];
	const bool in_lds = n <= MK_LDS;
	FinReg *f = in_lds ? s_f : a.fregs + rb0, *g = in_lds ? s_g : a.fregs2 + rb0;
	int *z = in_lds ? s_z : a.scr + 4 * rb0;                    // n ints (2n for select_records); in HBM they share the 4-int scratch of a region slot with the sort keys
	SortKey *keys = in_lds ? s_keys : reinterpret_cast<SortKey*>(a.scr + 4 * rb0);   // 16 B per region; z is taken after the sorts
	if (l == 0) { a.freg_n[r] = n; }
	if (n == 0) { if (l == 0) { a.n_pri[r] = 0; if (PLAN) { a.task_n[r] = 0; a.rec_n[r] = 0; } } return; }
	// id of region i for the tie-breaking hash (bwamem.c:534): SE n_processed + read; PE ((n_processed>>1) + pair)<<1 | end
	const uint64_t id = (opt.flag & BWAHIP_F_PE) ? ((((uint64_t)a.n_processed >> 1) + (uint64_t)(r >> 1)) << 1 | (uint64_t)(r & 1)) : (uint64_t)a.n_processed + (uint64_t)r;
	if (n == 1) {
		// a read with one region (most reads): what the code below does to it, written out -- it sorts to itself, overlaps nothing,
		// and ends with secondary = secondary_all = -1, sub = 0 whether it lies on an ALT contig or not (bwamem.c:528-565)
		if (l == 0) {
			const DevReg p = a.regs[rb0];
			FinReg q;
			q.rb = p.rb; q.re = p.re; q.hash = hash_64(id); q.frac_rep = p.frac_rep;
			q.qb = p.qb; q.qe = p.qe; q.rid = p.rid; q.score = p.score; q.truesc = p.truesc; q.sub = 0; q.alt_sc = 0; q.csub = p.csub;
			q.sub_n = p.sub_n; q.w = p.w; q.seedcov = p.seedcov; q.secondary = -1; q.secondary_all = -1; q.seedlen0 = p.seedlen0;
			q.n_comp = p.n_comp; q.is_alt = p.is_alt; q.pad = 0;
			a.fregs[rb0] = q;
			a.n_pri[r] = p.is_alt ? 0 : 1;
			if (PLAN) {
				const int rec = q.score >= opt.T ? 1 : 0;
				a.need[rb0] = (uint8_t)(rec ? NEED_REC : 0); a.xa_owner[rb0] = -1;
				a.task_n[r] = rec; a.rec_n[r] = rec;
			}
		}
		return;
	}
	int n_pri = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + l;
		bool pri = false;
		if (i < n) {
			const DevReg p = a.regs[rb0 + i];
			FinReg q;
			q.rb = p.rb; q.re = p.re; q.hash = hash_64(id + (uint64_t)i); q.frac_rep = p.frac_rep;
			q.qb = p.qb; q.qe = p.qe; q.rid = p.rid; q.score = p.score; q.truesc = p.truesc; q.sub = 0; q.alt_sc = 0; q.csub = p.csub;
			q.sub_n = p.sub_n; q.w = p.w; q.seedcov = p.seedcov; q.secondary = -1; q.secondary_all = -1; q.seedlen0 = p.seedlen0;
			q.n_comp = p.n_comp; q.is_alt = p.is_alt; q.pad = 0;
			g[i] = q;
			pri = !p.is_alt;
		}
		n_pri += __popcll(__ballot(pri));
	}
	wsync();
	rank_sort<0>(n, g, f, keys, l);                             // ks_introsort(mem_ars_hash), bwamem.c:537
	mark_core(opt, n, f, z, l);
	for (int i = l; i < n; i += 64) {                           // bwamem.c:539-544
		f[i].secondary_all = i;
		const int s = f[i].secondary;
		if (!f[i].is_alt && s >= 0 && f[s].is_alt) f[i].alt_sc = f[s].score;
	}
	wsync();
	if (n_pri >= 0 && n_pri < n) {                              // bwamem.c:545-558
		if (n_pri > 0) {
			rank_sort<1>(n, f, g, keys, l);                       // ks_introsort(mem_ars_hash2)
			for (int i = l; i < n; i += 64) f[i] = g[i];
			wsync();
		}
		for (int i = l; i < n; i += 64) z[f[i].secondary_all] = i;
		wsync();
		for (int i = l; i < n; i += 64) {
			if (f[i].secondary >= 0) { f[i].secondary_all = z[f[i].secondary]; if (f[i].is_alt) f[i].secondary = 0x7fffffff; }
			else f[i].secondary_all = -1;
		}
		wsync();
		if (n_pri > 0) {
			for (int i = l; i < n_pri; i += 64) { f[i].sub = 0; f[i].secondary = -1; }
			wsync();
			mark_core(opt, n_pri, f, z, l);
		}
	} else {
		for (int i = l; i < n; i += 64) f[i].secondary_all = f[i].secondary;
		wsync();
	}
	if (l == 0) a.n_pri[r] = n_pri;

	// ---- mem_reorder_primary5 (bwamem.c:988-1010), -5 (both the single-end path, bwamem.c:1205, and mem_sam_pe, bwamem_pair.c:305)
	if (opt.flag & BWAHIP_F_PRIMARY5) {
		if (l == 0) {
			int np = 0, left_st = 0x7fffffff, left_k = -1;
			for (int k = 0; k < n; ++k) if (f[k].secondary < 0 && !f[k].is_alt && f[k].score >= opt.T) ++np;
			if (np > 1) {
				for (int k = 0; k < n; ++k) {
					if (f[k].secondary >= 0 || f[k].is_alt || f[k].score < opt.T) continue;
					if (f[k].qb < left_st) { left_st = f[k].qb; left_k = k; }
				}
				if (left_k != 0) {
					{   // swap f[0] and f[left_k] 16 bytes at a time (a struct temporary would live in scratch memory)
						uint4 *x = reinterpret_cast<uint4*>(f), *y = reinterpret_cast<uint4*>(f + left_k);
						static_assert(sizeof(FinReg) == 96, "six 16-byte pieces");
#pragma unroll
						for (int q = 0; q < 6; ++q) { const uint4 tx = x[q], ty = y[q]; x[q] = ty; y[q] = tx; }
					}
					for (int k = 1; k < n; ++k) {
						if (f[k].secondary == 0) f[k].secondary = left_k; else if (f[k].secondary == left_k) f[k].secondary = 0;
						if (f[k].secondary_all == 0) f[k].secondary_all = left_k; else if (f[k].secondary_all == left_k) f[k].secondary_all = 0;
					}
				}
			}
		}
		wsync();
	}
	if (PLAN) {
		int n_task, n_rec;
		select_records(opt, n, f, a.need + rb0, a.xa_owner + rb0, z, l, n_task, n_rec);
		if (l == 0) { a.task_n[r] = n_task; a.rec_n[r] = n_rec; }
	}
	if (in_lds) {                                               // the marked regions to HBM, 16 bytes per lane and step
		wsync();
		const uint4 *src = reinterpret_cast<const uint4*>(s_f);
		uint4 *dst = reinterpret_cast<uint4*>(a.fregs + rb0);
		for (int i = l; i < n * 6; i += 64) dst[i] = src[i];
	}
}

__device__ __forceinline__ int infer_bw(int l1, int l2, int score, int a, int q, int r)   // bwamem.c:799
{
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = (int)(((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.));
	const int d = l1 > l2 ? l1 - l2 : l2 - l1;
	if (w < d) w = d;
	return w;
}
// the band mem_reg2aln starts with (bwamem.c:1118-1122); 0 with equal lengths means bwa_gen_cigar2 takes its no-DP shortcut
__device__ __forceinline__ int first_band(const DevOpt &opt, const FinReg &ar)
{
	const int lq = ar.qe - ar.qb, rlen = (int)(ar.re - ar.rb);
	const int tmp = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_del, opt.e_del);
	int w2 = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_ins, opt.e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
	return w2;
}

// the (read, region) pair of every alignment task, in read order then region order; and the two work lists of k_cigar: tasks
// that need no DP (equal lengths, band 0: a handful of mismatches at most -- four out of five at 1 % error) and the rest.
// List positions come from one block-wide scan and one atomic per block and list (1.2 M tasks on two counters otherwise).
__global__ __launch_bounds__(256) void k_task_fill(FinLaunch a)
{
	__shared__ int s_cnt[2][256];
	__shared__ int s_base[2];
	const int r = blockIdx.x * blockDim.x + threadIdx.x, tid = threadIdx.x;
	int n_fast = 0, n_dp = 0;
	const bool live = r < a.n_reads;
	const int n = live ? a.freg_n[r] : 0;
	const int64_t rb0 = live ? a.reg_base[r] : 0;
	int64_t t = live ? a.task_base[r] : 0;
	for (int i = 0; i < n; ++i) {
		if (a.need[rb0 + i]) {
			a.tasks[t] = make_int2(r, i); a.aln_of_reg[rb0 + i] = (int)t; ++t;
			const FinReg ar = a.fregs[rb0 + i];
			if (ar.qe - ar.qb == (int)(ar.re - ar.rb) && first_band(a.opt, ar) == 0) ++n_fast; else ++n_dp;
		} else a.aln_of_reg[rb0 + i] = -1;
	}
	s_cnt[0][tid] = n_fast; s_cnt[1][tid] = n_dp;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {                          // inclusive scans of both counts
		const int v0 = tid >= d ? s_cnt[0][tid - d] : 0, v1 = tid >= d ? s_cnt[1][tid - d] : 0;
		__syncthreads();
		s_cnt[0][tid] += v0; s_cnt[1][tid] += v1;
		__syncthreads();
	}
	if (tid == 255) { s_base[0] = atomicAdd(&a.list_n[0], s_cnt[0][255]); s_base[1] = atomicAdd(&a.list_n[1], s_cnt[1][255]); }
	__syncthreads();
	int pf = s_base[0] + s_cnt[0][tid] - n_fast, pd = s_base[1] + s_cnt[1][tid] - n_dp;
	t = live ? a.task_base[r] : 0;
	for (int i = 0; i < n; ++i) {
		if (!a.need[rb0 + i]) continue;
		const FinReg ar = a.fregs[rb0 + i];
		if (ar.qe - ar.qb == (int)(ar.re - ar.rb) && first_band(a.opt, ar) == 0) a.fast_list[pf++] = (int)t; else a.dp_list[pd++] = (int)t;
		++t;
	}
}

// ===================================================================================================
// K8 -- mem_reg2aln per selected region
// ===================================================================================================
constexpr int CG_MAXQ = BWAHIP_MAX_READ_LEN;
constexpr int CG_MAXT = 1536;                                // reference span of a region kept in LDS
constexpr int CG_ZLDS = 18432;                               // backtrack matrix bytes kept in LDS for reads of 192 bases and more (5120 below); 4 bits per cell in the band kernels
constexpr int CG_MAXC = 512;                                 // CIGAR operations staged in LDS
constexpr int CG_MAXMD = 1024;                               // MD bytes staged in LDS
constexpr unsigned long long CG_SLOT = 64;                   // bytes of pool every task owns (see reg2aln)
constexpr int CG_BIG_T = 8192;                               // k_cigar_big: reference span / matrix rows in its global slab
constexpr size_t CG_BIG_Z = (size_t)(CG_MAXQ + 1) * CG_BIG_T;

// ksw_global2 (ksw.c:504-584) with the backtrack matrix z (one byte per band cell, ksw.c:551-572).  Same column layout
// and max-plus scan for F as wave_global_score in k_extend.hip.  Returns the score.
template <int CPL>
__device__ int wave_global_trace(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen, int w, uint8_t *z, int n_col)
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
		int M[CPL], u[CPL], P = LOW;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			M[c] = Hs[c] + sw.mat[tb * 5 + qv[c]];
			u[c] = inb ? M[c] - oe_ins + j * e_ins : LOW;
			P = P > u[c] ? P : u[c];
		}
		int run = wscan_excl_max(P, LOW);
		int h[CPL];
		uint8_t *zi = z + (size_t)i * n_col;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int j = j0 + c;
			const bool inb = j >= beg && j < end;
			int f = NEG - (j - beg) * e_ins;
			const int g = run - (j - 1) * e_ins;
			if (j > beg && run > LOW) f = f > g ? f : g;
			int d = M[c] >= E[c] ? 0 : 1;
			int hv = M[c] >= E[c] ? M[c] : E[c];
			d = hv >= f ? d : 2;
			hv = hv >= f ? hv : f;
			h[c] = hv;
			if (inb) {
				const int tD = M[c] - oe_del;
				int en = E[c] - e_del;
				d |= en > tD ? 1 << 2 : 0;
				en = en > tD ? en : tD;
				E[c] = en;
				const int tI = M[c] - oe_ins;
				d |= f - e_ins > tI ? 2 << 4 : 0;
				zi[j - beg] = (uint8_t)d;
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
		if (end == beg) {
#pragma unroll
			for (int c = 0; c < CPL; ++c) if (j0 + c == end) Hs[c] = h1;
		}
	}
	int score = LOW;
#pragma unroll
	for (int c = 0; c < CPL; ++c) if (j0 + c == qlen) score = Hs[c];
	return wmax(score);
}


// ksw_global2 for a band of at most 64 diagonals (2w + 1 <= 64) -- what mem_reg2aln nearly always asks for (w = 3 .. 30 for a
// 150 bp read with a few mismatches or a short gap).  One lane per DIAGONAL k = j - i + w of the band, cells taken in
// anti-diagonal order: at step s lane k computes cell i = (s - k) / 2, j = i - w + k (every lane works every other step).
// Its three inputs are then exactly one step old: H(i-1,j-1) is the lane's own previous cell, E comes from lane k + 1's previous
// cell (i-1,j) and F from lane k - 1's previous cell (i,j-1) -- two DPP moves per step, no scan for F and no lanes idling
// outside the band as in the row-wise form.  Every cell evaluates ksw.c:548-572 literally, so H, E, F and the backtrack bytes are
// those of the reference whatever the order.  Needs w >= |tlen - qlen| (bwa.c:293-300 guarantees w >= |..| + 3): then no row is
// empty and the last row reaches column qlen - 1.  Returns the score (ksw.c:583).
// GZ: the backtrack cells go to a global slab instead (zg, 64 words per block of 16 steps: word (s >> 4) * 64 + k holds lane k's eight cells of
// those steps, 4 bits each) -- written with one coalesced store per 16 steps, so that the kernel's LDS holds nothing per cell and many more
// tasks are in flight; the backtrack reads it back run by run (reg2aln).
template <bool GZ>
__device__ int wave_band_trace(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen, int w, uint8_t *z, int n_col, unsigned *zg)
{
	const int k = lane();
	const int oe_del = sw.o_del + sw.e_del, oe_ins = sw.o_ins + sw.e_ins, e_del = sw.e_del, e_ins = sw.e_ins;
	// the cell before the first one of this diagonal lies on the border: row -1 (ksw.c:523-526) or column -1 (ksw.c:541)
	int Hd = k >= w ? (k == w ? 0 : -(sw.o_ins + e_ins * (k - w))) : -(sw.o_del + e_del * (w - k));
	int Eout = NEG, Fout = NEG;
	const int k_last = qlen - tlen + w;                          // diagonal of the corner cell (tlen-1, qlen-1)
	const int steps = 2 * (tlen - 1) + k_last;                   // its step; nothing after it matters
	int i = k >= w ? 0 : w - k, j = k >= w ? k - w : 0;          // first cell of this diagonal; every next one is (i+1, j+1)
	// the substitution score of a cell does not depend on the DP: it is looked up one cell ahead, and the two bases it needs two
	// cells ahead, so that no step waits on a chain of LDS round trips (indices clamped: values past the end are never used)
	auto ldt = [&](int ii) { return (int)t[(ii < tlen ? ii : tlen - 1) * ts]; };
	auto ldq = [&](int jj) { return (int)q[(jj < qlen ? jj : qlen - 1) * qs]; };
	int sc_next = sw.mat[ldt(i) * 5 + ldq(j)];
	int tn = ldt(i + 1), qn = ldq(j + 1);
	const bool mine = k <= 2 * w;
	int pend = 0;
	unsigned acc = 0;
	for (int s = 0; s <= steps; ++s) {
		const int Ein = __builtin_amdgcn_update_dpp(NEG, Eout, 0x130, 0xf, 0xf, false);   // wave_shl:1  lane k <- lane k + 1
		const int Fin = __builtin_amdgcn_update_dpp(NEG, Fout, 0x138, 0xf, 0xf, false);   // wave_shr:1  lane k <- lane k - 1
		if (mine && s - k == 2 * i && i < tlen && j < qlen) {
			const int m = Hd + sc_next;
			sc_next = sw.mat[tn * 5 + qn];
			tn = ldt(i + 2); qn = ldq(j + 2);
			int e = (i >= 1 && k < 2 * w) ? Ein : NEG;             // ksw.c:582 / 527: no cell above inside the band
			int f = (j >= 1 && k >= 1) ? Fin : NEG;                // ksw.c:538: f starts at MINUS_INF in every row
			int d = m >= e ? 0 : 1;
			int h = m >= e ? m : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			const int tD = m - oe_del;
			e -= e_del;
			d |= e > tD ? 1 << 2 : 0;
			Eout = e > tD ? e : tD;
			const int tI = m - oe_ins;
			f -= e_ins;
			d |= f > tI ? 2 << 4 : 0;
			Fout = f > tI ? f : tI;
			// 4 bits per cell (h source, e extended, f extended), rows 2r and 2r+1 of a diagonal in one byte at [r][k]: the lane writes the low
			// nibble with the even row and the whole byte with the odd one (half the LDS of a byte per cell: more tasks in flight)
			const int nib = (d & 3) | (d >> 2 & 1) << 2 | (d >> 5 & 1) << 3;
			if (GZ) acc |= (unsigned)nib << (((s & 15) >> 1) << 2);
			else if (i & 1) z[(size_t)(i >> 1) * n_col + k] = (uint8_t)(pend | nib << 4); else { pend = nib; z[(size_t)(i >> 1) * n_col + k] = (uint8_t)nib; }
			Hd = h;
			++i; ++j;
		}
		if (GZ && ((s & 15) == 15 || s == steps)) { zg[(size_t)(s >> 4) * 64 + k] = acc; acc = 0; }
	}
	return __builtin_amdgcn_readlane(Hd, k_last);
}

// The same for bands of up to 128 diagonals (2w + 1 <= 128): lane k owns the two diagonals 2k and 2k + 1.  Even steps take the cell
// of the even diagonal, odd steps the cell of the odd one -- both in row i = s/2 - k -- so every lane of the band works at every
// step.  Inputs, all one step old: E of (i-1, 2k+1) and F of (i, 2k) are the lane's own; F of (i, 2k-1) comes from lane k - 1, E of
// (i-1, 2k+2) from lane k + 1.
// GZ: as above, one byte (the lane's two cells of a row) per iteration, four iterations per word: word (it >> 2) * 64 + k, it = s / 2.
template <bool GZ>
__device__ int wave_band_trace2(const Sw &sw, const uint8_t *q, int qs, int qlen, const uint8_t *t, int ts, int tlen, int w, uint8_t *z, int n_col, unsigned *zg)
{
	const int k = lane(), d0 = 2 * k, d1 = d0 + 1;
	const int oe_del = sw.o_del + sw.e_del, oe_ins = sw.o_ins + sw.e_ins, e_del = sw.e_del, e_ins = sw.e_ins;
	auto border = [&](int d) { return d >= w ? (d == w ? 0 : -(sw.o_ins + e_ins * (d - w))) : -(sw.o_del + e_del * (w - d)); };   // ksw.c:523-526 / 541
	int H0 = border(d0), H1 = border(d1);
	int E0 = NEG, F0 = NEG, E1 = NEG, F1 = NEG;
	const int d_last = qlen - tlen + w;
	const int steps = 2 * (tlen - 1) + d_last;
	auto ldt = [&](int ii) { return (int)t[(ii < 0 ? 0 : ii < tlen ? ii : tlen - 1) * ts]; };
	auto ldq = [&](int jj) { return (int)q[(jj < 0 ? 0 : jj < qlen ? jj : qlen - 1) * qs]; };
	// scores one row ahead, bases two rows ahead (see wave_band_trace); row i of this lane's cells at iteration s is s/2 - k
	int i = -k, j0 = -k - w + d0;                                // row / column of the even cell at s = 0 (negative: not yet in the matrix)
	int tb1 = ldt(i), sc0 = sw.mat[tb1 * 5 + ldq(j0)], sc1 = sw.mat[tb1 * 5 + ldq(j0 + 1)];
	int tn = ldt(i + 1), qa = ldq(j0 + 1), qb = ldq(j0 + 2);
	int pend = 0;
	auto cell = [&](int d, int ii, int jj, int sc, int ein, int fin, int &Hd, int &Eo, int &Fo) {
		if (d <= 2 * w && ii >= 0 && ii < tlen && jj >= 0 && jj < qlen) {
			int e = (ii >= 1 && d < 2 * w) ? ein : NEG;
			int f = (jj >= 1 && d >= 1) ? fin : NEG;
			const int m = Hd + sc;
			int dd = m >= e ? 0 : 1;
			int h = m >= e ? m : e;
			dd = h >= f ? dd : 2;
			h = h >= f ? h : f;
			const int tD = m - oe_del;
			e -= e_del;
			dd |= e > tD ? 1 << 2 : 0;
			Eo = e > tD ? e : tD;
			const int tI = m - oe_ins;
			f -= e_ins;
			dd |= f > tI ? 2 << 4 : 0;
			Fo = f > tI ? f : tI;
			// 4 bits per cell, the lane's two diagonals of a row in one byte at [row][lane]: low nibble with the even diagonal, whole byte with the odd one
			const int nib = (dd & 3) | (dd >> 2 & 1) << 2 | (dd >> 5 & 1) << 3;
			if (GZ) pend |= (d & 1) ? nib << 4 : nib;
			else if (d & 1) z[(size_t)ii * n_col + k] = (uint8_t)(pend | nib << 4); else { pend = nib; z[(size_t)ii * n_col + k] = (uint8_t)nib; }
			Hd = h;
		}
	};
	unsigned acc = 0;
	for (int s = 0; s <= steps; s += 2) {
		pend = 0;
		const int c0 = sc0, c1 = sc1;
		sc0 = sw.mat[tn * 5 + qa]; sc1 = sw.mat[tn * 5 + qb];
		tn = ldt(i + 2); qa = ldq(j0 + 2); qb = ldq(j0 + 3);
		const int Fl = __builtin_amdgcn_update_dpp(NEG, F1, 0x138, 0xf, 0xf, false);   // wave_shr:1  F of lane k - 1's odd diagonal
		cell(d0, i, j0, c0, E1, Fl, H0, E0, F0);
		const int Er = __builtin_amdgcn_update_dpp(NEG, E0, 0x130, 0xf, 0xf, false);   // wave_shl:1  E of lane k + 1's even diagonal (its cell of row i - 1 ... computed this step)
		cell(d1, i, j0 + 1, c1, Er, F0, H1, E1, F1);
		++i; ++j0;
		if (GZ) {
			const int it = s >> 1;
			acc |= (unsigned)pend << ((it & 3) << 3);
			if ((it & 3) == 3 || s + 2 > steps) { zg[(size_t)(it >> 2) * 64 + k] = acc; acc = 0; }
		}
	}
	return __builtin_amdgcn_readlane((d_last & 1) ? H1 : H0, d_last >> 1);
}

// decimal digits of a non-negative integer into dst; returns the count
__device__ __forceinline__ int put_uint(uint8_t *dst, unsigned v)
{
	char b[12]; int n = 0;
	do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v);
	for (int i = 0; i < n; ++i) dst[i] = (uint8_t)b[n - 1 - i];
	return n;
}

struct CigarLds { uint8_t *q, *t, *z; uint32_t *cig; uint8_t *md; int8_t *mat; int max_c, max_md; unsigned *zg; };   // zg != nullptr: band kernels keep their backtrack cells in this global slab (k_cigar)   // max_c / max_md: capacity of cig / md

// One task: region `ar` of read r -> DevAln (+ CIGAR words and MD text appended to the pool).  BIG: window / matrix in the
// workgroup's global slab.  Returns false when the task does not fit this variant (caller lists it for k_cigar_big).
template <bool BIG, bool NODP = false, int CPLMAX = 11>
__device__ __forceinline__ bool reg2aln(const FinLaunch &a, const FinReg &ar, int r, long long task_id, const CigarLds &m, int t_cap, size_t z_cap, DevAln *out)
{
	const int l = lane();
	const DevOpt &opt = a.opt;
	const DevIndex &ix = a.ix;
	const int64_t l_pac = ix.l_pac;
	const int l_query = (int)(a.off[r + 1] - a.off[r]);
	const int qb = ar.qb, qe = ar.qe, lq = qe - qb;
	const int64_t rb = ar.rb, re = ar.re;
	const int rlen = (int)(re - rb);
	if (lq <= 0 || rb >= re || (rb < l_pac && re > l_pac)) { if (l == 0) atomicExch(a.err, 30); return true; }   // bwa_gen_cigar2 would return NULL: never for regions of mem_align1_core
	if (rlen > t_cap) return false;
	const bool rev = rb >= l_pac;
	int bad = 0;
	DevAln al;
	al.flag = ar.secondary >= 0 ? 0x100 : 0;
	al.mapq = ar.secondary < 0 ? (uint32_t)approx_mapq_se(opt, a.logtab, ar, &bad) & 0xff : 0;
	if (bad && l == 0) atomicExch(a.err, 31);
	// reference span, one base per byte; for a reverse-strand hit query and reference are both read backwards (bwa.c:275-280)
	__syncthreads();
	for (int i = l; i < rlen; i += 64) m.t[i] = (uint8_t)ref_base(ix, rb + i);
	__syncthreads();
	const uint8_t *qp = rev ? m.q + qe - 1 : m.q + qb; const int qs = rev ? -1 : 1;
	const uint8_t *tp = rev ? m.t + rlen - 1 : m.t; const int ts = rev ? -1 : 1;
	Sw sw; sw.mat = m.mat; sw.o_del = opt.o_del; sw.e_del = opt.e_del; sw.o_ins = opt.o_ins; sw.e_ins = opt.e_ins; sw.mx = 0;
	int w2;
	{
		const int tmp = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_del, opt.e_del);
		w2 = infer_bw(lq, rlen, ar.truesc, opt.a, opt.o_ins, opt.e_ins);
		w2 = w2 > tmp ? w2 : tmp;
		if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
	}
	int score = 0, last_sc = -(1 << 30), n_cigar = 0, it = 0;
	bool fits = true;
	do {                                                         // bwamem.c:1124-1132
		w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
		// ---- bwa_gen_cigar2 (bwa.c:281-307)
		if (NODP || (lq == rlen && w2 == 0)) {                    // NODP: the caller established that the shortcut is taken (k_task_fill)
			int part = 0;
			for (int i = l; i < lq; i += 64) part += m.mat[tp[i * ts] * 5 + qp[i * qs]];
			score = wsum(part);
			n_cigar = 1;
			if (l == 0) m.cig[0] = (uint32_t)lq << 4 | 0;
		} else {
			int max_ins = div_plus1_trunc(((lq + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins);
			int max_del = div_plus1_trunc(((lq + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del);
			int max_gap = max_ins > max_del ? max_ins : max_del;
			max_gap = max_gap > 1 ? max_gap : 1;
			int dl = rlen - lq; dl = dl < 0 ? -dl : dl;
			int w = (max_gap + dl + 1) >> 1;
			w = w < w2 ? w : w2;
			const int min_w = dl + 3;
			w = w > min_w ? w : min_w;
			const int n_col = lq < 2 * w + 1 ? lq : 2 * w + 1;
			const bool nib_z = 2 * w + 1 <= 64;                     // band of at most 64 diagonals: 4-bit cells, two rows per byte (wave_band_trace)
			const bool nib2_z = !nib_z && 2 * w + 1 <= 128;          // up to 128: 4-bit cells, a lane's two diagonals per byte (wave_band_trace2)
			const bool gz = !BIG && m.zg != nullptr;                   // k_cigar: band cells in the global slab, nothing wider than 128 diagonals
			if (gz ? !(nib_z || nib2_z) : (nib_z ? (size_t)((rlen + 1) >> 1) * (size_t)(2 * w + 1) : nib2_z ? (size_t)rlen * (size_t)(w + 1) : (size_t)n_col * (size_t)rlen) > z_cap) { fits = false; break; }
			__syncthreads();
			// the fewest columns per lane that hold the query; CPLMAX (from the longest read of the batch) bounds what is
			// compiled in, and with it the registers of the kernel
			if (!BIG && gz) score = nib_z ? wave_band_trace<true>(sw, qp, qs, lq, tp, ts, rlen, w, nullptr, 0, m.zg) : wave_band_trace2<true>(sw, qp, qs, lq, tp, ts, rlen, w, nullptr, 0, m.zg);
			else if (nib_z) score = wave_band_trace<false>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, 2 * w + 1, nullptr);   // (row stride = diagonals of the band)
			else if (nib2_z) score = wave_band_trace2<false>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, w + 1, nullptr);   // (row stride = lanes of the band)
			else if (BIG && lq < 64) score = wave_global_trace<1>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, n_col);
			else if (BIG && (CPLMAX <= 2 || lq < 128)) score = wave_global_trace<(CPLMAX < 2 ? CPLMAX : 2)>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, n_col);
			else if (BIG && (CPLMAX <= 3 || lq < 192)) score = wave_global_trace<(CPLMAX < 3 ? CPLMAX : 3)>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, n_col);
			else if (BIG && (CPLMAX <= 4 || lq < 256)) score = wave_global_trace<(CPLMAX < 4 ? CPLMAX : 4)>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, n_col);
			else if (BIG) score = wave_global_trace<CPLMAX>(sw, qp, qs, lq, tp, ts, rlen, w, m.z, n_col);
			wsync();
			// ---- backtrack (ksw.c:586-603); operations are produced last to first and reversed afterwards.  The walk is a chain of dependent
			// reads of z, but nearly all of its steps are diagonal ones in state 0 (match / mismatch): there the next cells are known in
			// advance -- (i - t, k - t) -- so 64 of them are read at once, one per lane, and a ballot finds the first that leaves the
			// diagonal.  Gap states (1: deletion, 2: insertion) are followed cell by cell.  Same cells, same decisions, same operations.
			int nc = 0;
			{
				int which = 0, i = rlen - 1, k = (i + w + 1 < lq ? i + w + 1 : lq) - 1;
				uint32_t cur = 0; bool have = false, ovf = false;
				auto push = [&](int op, int len) {                    // uniform over the wavefront; lane 0 stores
					if (have && (int)(cur & 0xf) == op) cur += (uint32_t)len << 4;
					else { if (have) { if (nc < m.max_c) { if (l == 0) m.cig[nc] = cur; } else ovf = true; ++nc; } cur = (uint32_t)len << 4 | (uint32_t)op; have = true; }
				};
				auto cell = [&](int ii, int kk) -> int {              // the z byte of ksw.c:551-572 for cell (row ii, column kk)
					if (gz) {
						const int kd = kk - ii + w;
						int nb;
						if (nib_z) { const int st = 2 * ii + kd; nb = (int)(m.zg[(size_t)(st >> 4) * 64 + kd] >> (((st & 15) >> 1) << 2)) & 15; }
						else { const int kl = kd >> 1, it = ii + kl; const int byte = (int)(m.zg[(size_t)(it >> 2) * 64 + kl] >> ((it & 3) << 3)) & 255; nb = (kd & 1) ? byte >> 4 : byte & 15; }
						return (nb & 3) | (nb >> 2 & 1) << 2 | (nb >> 3 & 1) << 5;
					}
					if (nib_z || nib2_z) {
						const int kd = kk - ii + w;
						const int byte = nib_z ? m.z[(size_t)(ii >> 1) * (2 * w + 1) + kd] : m.z[(size_t)ii * (w + 1) + (kd >> 1)];
						const int nb = (nib_z ? (ii & 1) : (kd & 1)) ? byte >> 4 : byte & 15;
						return (nb & 3) | (nb >> 2 & 1) << 2 | (nb >> 3 & 1) << 5;
					}
					return m.z[(size_t)ii * n_col + (kk - (ii > w ? ii - w : 0))];
				};
				while (i >= 0 && k >= 0) {
					if (which == 0) {
						const int ii = i - l, kk = k - l;
						const bool valid = ii >= 0 && kk >= 0;
						const int d = valid ? cell(ii, kk) : 0;
						const unsigned long long stop = __ballot(!valid || (d & 3) != 0);
						const int run = stop ? __ffsll((long long)stop) - 1 : 64;
						if (run > 0) { push(0, run); i -= run; k -= run; }
						if (stop) {
							const int dr = __shfl(valid ? d : 0, run);        // 0: the matrix ended there (i or k is negative now)
							which = dr & 3;
							if (which == 1) { push(2, 1); --i; } else if (which == 2) { push(1, 1); --k; }
						}
					} else {
						which = cell(i, k) >> (which << 1) & 3;
						if (which == 0) { push(0, 1); --i; --k; }
						else if (which == 1) { push(2, 1); --i; }
						else { push(1, 1); --k; }
					}
				}
				if (i >= 0) push(2, i + 1);
				if (k >= 0) push(1, k + 1);
				if (have) { if (nc < m.max_c) { if (l == 0) m.cig[nc] = cur; } else ovf = true; ++nc; }
				wsync();
				if (ovf) nc = -1;
				else if (l == 0) for (int x = 0; x < nc >> 1; ++x) { const uint32_t tmp = m.cig[x]; m.cig[x] = m.cig[nc - 1 - x]; m.cig[nc - 1 - x] = tmp; }
			}
			nc = __shfl(nc, 0);
			if (nc < 0) { fits = false; break; }
			n_cigar = nc;
			wsync();
		}
		if (score == last_sc || w2 == opt.w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++it < 3 && score < ar.truesc - opt.a);
	if (!fits) return false;

	// ---- NM and MD (bwa.c:309-339) on the raw CIGAR; for a reverse-strand hit the bases come out complemented ("TGCAN")
	int md_len = 0, NM = 0;
	{
		int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0;
		bool md_ovf = false;
		auto emit_num = [&](int v) { if (md_len + 11 < m.max_md) { if (l == 0) md_len += put_uint(m.md + md_len, (unsigned)v); } else md_ovf = true; };
		// md_len is advanced by lane 0 only; it is broadcast after every variable-length emission
		const char *int2base = rev ? "TGCAN" : "ACGTN";
		for (int k = 0; k < n_cigar; ++k) {
			const uint32_t cg = m.cig[k];
			const int op = cg & 0xf, len = (int)(cg >> 4);
			if (op == 0) {
				for (int base = 0; base < len; base += 64) {
					const int i = base + l;
					const bool mm = i < len && qp[(x + i) * qs] != tp[(y + i) * ts];
					unsigned long long mask = __ballot(mm);
					int prev = 0;
					while (mask) {
						const int p = __ffsll((long long)mask) - 1;
						mask &= mask - 1;
						u += p - prev;
						emit_num(u);
						md_len = __shfl(md_len, 0);
						if (md_len + 1 < m.max_md) { if (l == 0) m.md[md_len] = (uint8_t)int2base[tp[(y + base + p) * ts]]; ++md_len; } else md_ovf = true;
						++n_mm; u = 0; prev = p + 1;
					}
					const int chunk = len - base < 64 ? len - base : 64;
					u += chunk - prev;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < n_cigar - 1) {
					emit_num(u);
					md_len = __shfl(md_len, 0);
					if (md_len + 1 + len < m.max_md) {
						if (l == 0) m.md[md_len] = '^';
						for (int i = l; i < len; i += 64) m.md[md_len + 1 + i] = (uint8_t)int2base[tp[(y + i) * ts]];
						md_len += 1 + len;
					} else md_ovf = true;
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) { x += len; n_gap += len; }
		}
		emit_num(u);
		md_len = __shfl(md_len, 0);
		NM = n_mm + n_gap;
		if (md_ovf) return false;
	}
	wsync();
	al.NM = (uint32_t)NM & 0x3fffff;
	// ---- position, end deletions, clips (bwamem.c:1134-1168)
	const int is_rev = rev ? 1 : 0;
	int64_t pos = rev ? (l_pac << 1) - 1 - (re - 1) : rb;         // bns_depos(rb < l_pac ? rb : re - 1)
	int c0 = 0, c1 = n_cigar;                                     // kept CIGAR words [c0, c1)
	if (n_cigar > 0) {
		if ((m.cig[0] & 0xf) == 2) { pos += m.cig[0] >> 4; c0 = 1; }
		else if ((m.cig[n_cigar - 1] & 0xf) == 2) c1 = n_cigar - 1;
	}
	int clip5 = 0, clip3 = 0;
	if (qb != 0 || qe != l_query) { clip5 = is_rev ? l_query - qe : qb; clip3 = is_rev ? qb : l_query - qe; }
	const int n_out = (c1 - c0) + (clip5 ? 1 : 0) + (clip3 ? 1 : 0);
	al.is_rev = (uint32_t)is_rev;
	al.rid = dev_pos2rid(ix, pos);
	al.pos = pos - ix.anns[al.rid].offset;
	al.score = ar.score; al.sub = ar.sub > ar.csub ? ar.sub : ar.csub;
	al.is_alt = (uint32_t)ar.is_alt; al.alt_sc = ar.alt_sc;
	al.n_cigar = n_out; al.md_len = md_len; al.pad = 0; al.pad2 = 0;
	// ---- CIGAR words + MD text into the pool
	// every task owns a 64-byte slot at the start of the pool (CIGAR words + MD of an ordinary hit fit); only longer texts take
	// space from the shared tail through an atomic (1.2 M tasks bumping one counter cost more than everything else here)
	const unsigned long long bytes = ((unsigned long long)n_out * 4 + (unsigned long long)md_len + 7) & ~7ull;
	unsigned long long at = (unsigned long long)task_id * CG_SLOT;
	if (bytes > CG_SLOT) {
		if (l == 0) at = atomicAdd(a.pool_head, bytes);
		at = (unsigned long long)__shfl((long long)at, 0);
		if (at + bytes > a.pool_cap) { if (l == 0) atomicExch(a.err, 5); return true; }   // pool exhausted: the host re-runs the stage with a larger one
	}
	uint32_t *dc = reinterpret_cast<uint32_t*>(a.pool + at);
	const int lead = clip5 ? 1 : 0;
	if (l == 0) { if (clip5) dc[0] = (uint32_t)clip5 << 4 | 3; if (clip3) dc[n_out - 1] = (uint32_t)clip3 << 4 | 3; }
	for (int i = l; i < c1 - c0; i += 64) dc[lead + i] = m.cig[c0 + i];
	uint8_t *dm = a.pool + at + (size_t)n_out * 4;
	for (int i = l; i < md_len; i += 64) dm[i] = m.md[i];
	al.cigar_off = (int64_t)at; al.md_off = (int64_t)(at + (unsigned long long)n_out * 4);
	if (l == 0) *out = al;
	return true;
}

// FAST: the tasks of fast_list (no DP: no backtrack matrix, one CIGAR operation), one task per workgroup.  DP tasks (FAST = false): a fixed
// grid of workgroups takes them in turn; each owns a slab of global memory for the backtrack cells of the band kernels (written with
// coalesced stores, read back run by run), so that LDS holds only the query, the reference span and the CIGAR / MD being built and the
// number of tasks in flight is set by registers.  Bands wider than 128 diagonals, or texts beyond the staging arrays, go to k_cigar_big.
constexpr int CG_TCAP_SMALL = 768;
// wavefronts per SIMD k_cigar is compiled for: 6 (80 VGPRs) for reads below 192 bases, 5 (96) above -- measured: 5.9 vs 6.2 ms (150 bp) and
// 44 vs 47 ms (250 bp at 5 %) per million reads; 8 (64 VGPRs, spilling) loses in both
__host__ __device__ constexpr size_t cigar_zslab_words(int tcap) { return (size_t)(tcap + 72) * 16; }   // covers both band kernels: (tlen + w + 4) / 4 and (2 tlen + 2 w) / 16 blocks of 64 words
template <bool FAST, int CPLMAX>
__global__ __launch_bounds__(64, FAST ? 7 : CPLMAX <= 3 ? 6 : 5) void k_cigar(FinLaunch a, int n_list)
{
	constexpr bool SMALL = FAST || CPLMAX <= 3;
	constexpr int TCAP = SMALL ? CG_TCAP_SMALL : CG_MAXT, MC = FAST ? 8 : SMALL ? 160 : CG_MAXC, MMD = SMALL ? 512 : CG_MAXMD;
	__shared__ uint8_t s_q[CG_MAXQ + 8];
	__shared__ uint8_t s_t[TCAP + 8];
	__shared__ uint8_t s_z[16];
	__shared__ uint32_t s_cig[MC];
	__shared__ uint8_t s_md[MMD];
	__shared__ int8_t s_mat[32];
	const int l = lane();
	if (l < 25) s_mat[l] = a.opt.mat[l];
	unsigned *zg = FAST ? nullptr : a.zslab + (size_t)blockIdx.x * cigar_zslab_words(TCAP);
	for (int it = (int)blockIdx.x; it < n_list; it += FAST ? n_list : (int)gridDim.x) {   // (FAST: one task per workgroup, the grid is the list)
		const long long t = FAST ? a.fast_list[it] : a.dp_list[it];
		const int2 tk = a.tasks[t];
		const int r = tk.x;
		const int l_query = (int)(a.off[r + 1] - a.off[r]);
		const uint8_t *query = a.seq + a.off[r];
		__syncthreads();
		for (int i = l; i < l_query; i += 64) { const uint8_t c = query[i]; s_q[i] = c < 5 ? c : 4; }   // bwamem.c:1115: codes >= 5 -> 4 (already codes here)
		__syncthreads();
		const FinReg ar = a.fregs[a.reg_base[r] + tk.y];
		const CigarLds m = { s_q, s_t, s_z, s_cig, s_md, s_mat, MC, MMD, zg };
		if (!reg2aln<false, FAST, CPLMAX>(a, ar, r, t, m, TCAP, 0, a.alns + t) && l == 0) a.redo_list[atomicAdd(a.redo_n, 1)] = (int)t;
	}
}

// tasks whose reference span, backtrack matrix, CIGAR or MD did not fit LDS: the same code on this workgroup's global slab
__global__ __launch_bounds__(64) void k_cigar_big(FinLaunch a)
{
	__shared__ uint8_t s_q[CG_MAXQ + 8];
	__shared__ uint32_t s_cig[CG_MAXC];
	__shared__ uint8_t s_md[CG_MAXMD];
	__shared__ int8_t s_mat[32];
	const int l = lane();
	uint8_t *slab = a.big_z + (size_t)blockIdx.x * (CG_BIG_Z + CG_BIG_T + 64);
	const int n_redo = *a.redo_n;
	for (int it = (int)blockIdx.x; it < n_redo; it += (int)gridDim.x) {
		const long long t = a.redo_list[it];
		const int2 tk = a.tasks[t];
		const int r = tk.x;
		const int l_query = (int)(a.off[r + 1] - a.off[r]);
		const uint8_t *query = a.seq + a.off[r];
		__syncthreads();
		for (int i = l; i < l_query; i += 64) { const uint8_t c = query[i]; s_q[i] = c < 5 ? c : 4; }
		if (l < 25) s_mat[l] = a.opt.mat[l];
		__syncthreads();
		const FinReg ar = a.fregs[a.reg_base[r] + tk.y];
		const CigarLds m = { s_q, slab + CG_BIG_Z, slab, s_cig, s_md, s_mat, CG_MAXC, CG_MAXMD };
		if (!reg2aln<true>(a, ar, r, t, m, CG_BIG_T, CG_BIG_Z, a.alns + t) && l == 0) { atomicExch(a.err, 6); atomicExch(a.err + 1, r); }
	}
}

} // namespace

int launch_mark_primary(const FinLaunch &a, bool plan, int n_listed, hipStream_t st)
{
	const int grid = a.subset == 2 ? 2 * n_listed : a.n_reads;
	if (grid <= 0) return 0;
	if (plan) hipLaunchKernelGGL(k_mark<true>, dim3(grid), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_mark<false>, dim3(grid), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_task_fill(const FinLaunch &a, hipStream_t st)
{
	if (a.n_reads <= 0) return 0;
	hipLaunchKernelGGL(k_task_fill, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

size_t cigar_big_slab_bytes() { return CG_BIG_Z + CG_BIG_T + 64; }

// the two lists run side by side (st2 forks from st and joins it again): the no-DP tasks wait on memory, the DP tasks compute
// grid of the DP kernel and the bytes of backtrack slabs it needs (run_final sizes FinLaunch::zslab with it)
int cigar_dp_grid(int n_dp) { return n_dp < 256 * 28 ? n_dp : 256 * 28; }
size_t cigar_zslab_bytes(int max_len, int n_dp)
{
	return (size_t)cigar_dp_grid(n_dp) * cigar_zslab_words(max_len < 64 * 3 ? CG_TCAP_SMALL : CG_MAXT) * 4;
}
int launch_cigar(const FinLaunch &a, int n_fast, int n_dp, int max_len, hipStream_t st, hipStream_t st2, hipEvent_t fork, hipEvent_t join)
{
	const bool both = n_fast > 0 && n_dp > 0;
	if (both && (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(st2, fork, 0) != hipSuccess)) return BWAHIP_ENODEV;
	if (n_dp > 0) {                                             // query columns of the longest read over 64 lanes
		const unsigned grid = (unsigned)cigar_dp_grid(n_dp);
		if (max_len < 64 * 3) hipLaunchKernelGGL((k_cigar<false, 3>), dim3(grid), dim3(64), 0, st, a, n_dp);
		else if (max_len < 64 * 5) hipLaunchKernelGGL((k_cigar<false, 5>), dim3(grid), dim3(64), 0, st, a, n_dp);
		else hipLaunchKernelGGL((k_cigar<false, 11>), dim3(grid), dim3(64), 0, st, a, n_dp);
	}
	if (n_fast > 0) hipLaunchKernelGGL((k_cigar<true, 1>), dim3((unsigned)n_fast), dim3(64), 0, both ? st2 : st, a, n_fast);
	if (both && (hipEventRecord(join, st2) != hipSuccess || hipStreamWaitEvent(st, join, 0) != hipSuccess)) return BWAHIP_ENODEV;
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_cigar_big(const FinLaunch &a, int grid, hipStream_t st)
{
	hipLaunchKernelGGL(k_cigar_big, dim3(grid), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
