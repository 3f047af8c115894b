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

namespace {

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void wsync() { __threadfence_block(); __syncthreads(); }
__device__ __forceinline__ int wsum(int v) { for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d); return v; }

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

// mem_mark_primary_se_core (bwamem.c:500-526) on f[0..n): z = list of kept (non-secondary) regions.  Region i is tested
// against the kept ones, one per lane; the reference stops at the FIRST kept region it overlaps significantly.
__device__ void mark_core(const DevOpt &opt, int n, FinReg *f, int *z, int l)
{
	int tmp = opt.a + opt.b;
	tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
	tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
	if (n <= 0) return;
	if (l == 0) z[0] = 0;
	int nz = 1;
	wsync();
	for (int i = 1; i < n; ++i) {
		const int iqb = f[i].qb, iqe = f[i].qe, isc = f[i].score, ialt = f[i].is_alt;
		int hit = -1;
		for (int base = 0; base < nz && hit < 0; base += 64) {
			const int k = base + l;
			bool sig = false;
			if (k < nz) {
				const int j = z[k];
				const int jqb = f[j].qb, jqe = f[j].qe;
				const int b_max = jqb > iqb ? jqb : iqb, e_min = jqe < iqe ? jqe : iqe;
				if (e_min > b_max) {
					const int min_l = iqe - iqb < jqe - jqb ? iqe - iqb : jqe - jqb;
					if ((float)(e_min - b_max) >= (float)min_l * opt.mask_level) sig = true;
				}
			}
			const unsigned long long m = __ballot(sig);
			if (m) hit = base + __ffsll((long long)m) - 1;
		}
		if (hit >= 0) {
			if (l == 0) {
				const int j = z[hit];
				if (f[j].sub == 0) f[j].sub = isc;
				if (f[j].score - isc <= tmp && (f[j].is_alt || !ialt)) ++f[j].sub_n;
				f[i].secondary = j;
			}
		} else {
			if (l == 0) z[nz] = i;
			++nz;
		}
		wsync();
	}
}

// rank sort of f[0..n) into g[0..n) by MODE's order (keys unique); keys in `keys`
template <int MODE> __device__ void rank_sort(int n, const FinReg *f, FinReg *g, SortKey *keys, int l)
{
	for (int i = l; i < n; i += 64) { keys[i].score = f[i].score; keys[i].is_alt = f[i].is_alt; keys[i].hash = f[i].hash; }
	wsync();
	for (int i = l; i < n; i += 64) {
		const SortKey ki = keys[i];
		int rank = 0;
		for (int j = 0; j < n; ++j) rank += key_lt<MODE>(keys[j], ki) ? 1 : 0;
		g[rank] = f[i];
	}
	wsync();
}

// One read per wavefront.  PLAN: also select records / XA members (single-end output path).
template <bool PLAN>
__global__ __launch_bounds__(64) void k_mark(FinLaunch a)
{
	const int r = blockIdx.x, l = lane();
	const DevOpt &opt = a.opt;
	const int n = a.reg_n[r];
	const int64_t rb0 = a.reg_base[r];
	FinReg *f = a.fregs + rb0, *g = a.fregs2 + rb0;
	int *z = a.scr + 4 * rb0;                                   // n ints; the next 3n ints: sort keys (16 B each would need 4n: keys live in g's slots' tail instead)
	SortKey *keys = reinterpret_cast<SortKey*>(a.scr + 4 * rb0);   // 16 B per region = the whole 4-int scratch; z is taken after the sorts
	if (l == 0) { a.freg_n[r] = n; }
	if (n == 0) { if (l == 0) { a.n_pri[r] = 0; if (PLAN) { a.task_n[r] = 0; a.rec_n[r] = 0; } } return; }
	// id of region i for the tie-breaking hash (bwamem.c:534): SE n_processed + read; PE ((n_processed>>1) + pair)<<1 | end
	const uint64_t id = (opt.flag & BWAHIP_F_PE) ? ((((uint64_t)a.n_processed >> 1) + (uint64_t)(r >> 1)) << 1 | (uint64_t)(r & 1)) : (uint64_t)a.n_processed + (uint64_t)r;
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
	if (!PLAN) return;

	// ---- mem_reorder_primary5 (bwamem.c:988-1010), -5
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
					const FinReg t = f[0]; f[0] = f[left_k]; f[left_k] = t;
					for (int k = 1; k < n; ++k) {
						if (f[k].secondary == 0) f[k].secondary = left_k; else if (f[k].secondary == left_k) f[k].secondary = 0;
						if (f[k].secondary_all == 0) f[k].secondary_all = left_k; else if (f[k].secondary_all == left_k) f[k].secondary_all = 0;
					}
				}
			}
		}
		wsync();
	}

	// ---- selection: mem_gen_alt's XA membership (bwamem_extra.c:116-145) and mem_reg2sam's record filter (bwamem.c:1025-1031)
	uint8_t *need = a.need + rb0;
	int *owner = a.xa_owner + rb0;
	int *cnt = z, *has_alt = z + n;                             // z is free now (2n of the 4n scratch ints)
	for (int i = l; i < n; i += 64) { cnt[i] = 0; has_alt[i] = 0; need[i] = 0; owner[i] = -1; }
	wsync();
	const bool want_xa = !(opt.flag & BWAHIP_F_ALL);
	if (want_xa) {
		for (int i = l; i < n; i += 64) {
			const int k = f[i].secondary_all;
			int pr = -1;
			if (k >= 0 && (double)f[i].score >= (double)f[k].score * (double)opt.XA_drop_ratio) pr = k;   // get_pri_idx: int >= int * double
			owner[i] = pr;
			if (pr >= 0) { atomicAdd(&cnt[pr], 1); if (f[i].is_alt) atomicOr(&has_alt[pr], 1); }
		}
		wsync();
	}
	int n_task = 0, n_rec = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + l;
		int nd = 0;
		if (i < n) {
			const FinReg p = f[i];
			bool rec = p.score >= opt.T;
			if (rec && p.secondary >= 0 && (p.is_alt || !(opt.flag & BWAHIP_F_ALL))) rec = false;
			if (rec && p.secondary >= 0 && p.secondary < 0x7fffffff && (float)p.score < (float)f[p.secondary].score * opt.drop_ratio) rec = false;
			if (rec) nd |= NEED_REC;
			const int pr = owner[i];
			if (want_xa && pr >= 0 && !(cnt[pr] > opt.max_XA_hits_alt || (!has_alt[pr] && cnt[pr] > opt.max_XA_hits))) nd |= NEED_XA;
			else owner[i] = -1;
			need[i] = (uint8_t)nd;
		}
		n_task += __popcll(__ballot(nd != 0));
		n_rec += __popcll(__ballot((nd & NEED_REC) != 0));
	}
	if (l == 0) { a.task_n[r] = n_task; a.rec_n[r] = n_rec; }
}

// the (read, region) pair of every alignment task, in read order then region order
__global__ void k_task_fill(FinLaunch a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.n_reads) return;
	const int n = a.freg_n[r];
	const int64_t rb0 = a.reg_base[r];
	int64_t t = a.task_base[r];
	for (int i = 0; i < n; ++i) {
		if (a.need[rb0 + i]) { a.tasks[t] = make_int2(r, i); a.aln_of_reg[rb0 + i] = (int)t; ++t; }
		else a.aln_of_reg[rb0 + i] = -1;
	}
}

} // namespace

int launch_mark_primary(const FinLaunch &a, bool plan, hipStream_t st)
{
	if (a.n_reads <= 0) return 0;
	if (plan) hipLaunchKernelGGL(k_mark<true>, dim3(a.n_reads), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_mark<false>, dim3(a.n_reads), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_task_fill(const FinLaunch &a, hipStream_t st)
{
	if (a.n_reads <= 0) return 0;
	hipLaunchKernelGGL(k_task_fill, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
