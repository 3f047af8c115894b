// Pieces of the finalisation shared by the single-end path (k_final.hip) and the paired-end path (k_pair.hip).
#pragma once
#include "bwahip_internal.h"

namespace fin {

__device__ __forceinline__ void wsync() { __threadfence_block(); __syncthreads(); }

// mem_approx_mapq_se (bwamem.c:962-986); log() of integers from the host-made table
__device__ int approx_mapq_se(const DevOpt &o, const double *logtab, const FinReg &a, int *bad)
{
	int mapq, l, sub = a.sub ? a.sub : o.min_seed_len * o.a;
	double identity;
	sub = a.csub > sub ? a.csub : sub;
	if (sub >= a.score) return 0;
	l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
	identity = 1. - (double)(l * o.a - a.score) / (o.a + o.b) / l;
	if (l >= BWAHIP_LOGTAB_N || a.sub_n + 1 >= BWAHIP_LOGTAB_N || a.seedcov >= BWAHIP_LOGTAB_N || l < 1) { *bad = 1; return 0; }
	if (a.score == 0) mapq = 0;
	else if (o.mapQ_coef_len > 0) {
		double tmp = (float)l < o.mapQ_coef_len ? 1. : (double)o.mapQ_coef_fac / logtab[l];
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a.score - sub) / o.a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a.score) * logtab[a.seedcov > 0 ? a.seedcov : 0] + .499);   // seedcov == 0: log(0) = -inf in the reference; not reachable (a region covers its seed)
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a.sub_n > 0) mapq -= (int)(4.343 * logtab[a.sub_n + 1] + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a.frac_rep) + .499);
	return mapq;
}


// The selection mem_reg2sam (bwamem.c:1025-1031) and mem_gen_alt (bwamem_extra.c:116-145) make on a read's marked
// regions f[0..n): need[i] |= NEED_REC for regions that print a record, NEED_XA (+ owner[i] = the record's region) for
// the ones listed in an XA tag.  z: 2n ints of scratch.  Wavefront-collective; counts are returned in every lane.
__device__ void select_records(const DevOpt &opt, int n, const FinReg *f, uint8_t *need, int *owner, int *z, int l, int &n_task, int &n_rec)
{
	if (n == 1) {                                               // the common case, without the list machinery: a lone region is nobody's secondary
		const int rec = f[0].score >= opt.T ? 1 : 0;
		if (l == 0) { need[0] = (uint8_t)(rec ? NEED_REC : 0); owner[0] = -1; }
		n_task = n_rec = rec;
		return;
	}
	// ---- selection: mem_gen_alt's XA membership (bwamem_extra.c:116-145) and mem_reg2sam's record filter (bwamem.c:1025-1031)
	int *cnt = z, *has_alt = z + n;                             // z is free now (2n of the 4n scratch ints)
	for (int i = l; i < n; i += 64) { cnt[i] = 0; has_alt[i] = 0; need[i] = 0; owner[i] = -1; }
	wsync();
	const bool want_xa = !(opt.flag & BWAHIP_F_ALL);
	if (want_xa) {
		for (int base = 0; base < n; base += 64) {
			const int i = base + l;
			int pr = -1, alt = 0;
			if (i < n) {
				const int k = f[i].secondary_all;
				if (k >= 0 && (double)f[i].score >= (double)f[k].score * (double)opt.XA_drop_ratio) pr = k;   // get_pri_idx: int >= int * double
				owner[i] = pr;
				alt = f[i].is_alt;
			}
			// counts per primary, one update per distinct primary among the 64 lanes (a read inside a repeat family has hundreds of hits
			// under ONE primary: an atomic per hit would queue them all on one address)
			unsigned long long todo = __ballot(pr >= 0);
			while (todo) {
				const int lead = __ffsll((long long)todo) - 1;
				const int p0 = __shfl(pr, lead);
				const unsigned long long same = __ballot(pr == p0);
				const unsigned long long any_alt = __ballot(pr == p0 && alt);
				if (l == lead) { cnt[p0] += __popcll(same); if (any_alt) has_alt[p0] = 1; }
				todo &= ~same;
			}
		}
		wsync();
	}
	n_task = 0; n_rec = 0;
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
}

} // namespace fin
