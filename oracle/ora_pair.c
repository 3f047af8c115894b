/* ORACLE (test infrastructure only) -- paired-end logic.
 * Restates bwamem_pair.c: insert-size statistics (:72-135), mate rescue
 * (:137-206), pairing (:208-271), paired SAM output and mapQ (:276-419).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "ora.h"

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)   /* bwamem_pair.c:48 */
{
	int64_t p2;
	int r1 = (b1 >= l_pac), r2 = (b2 >= l_pac);
	p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

static int best_sub(const ora_opt_t *opt, const ora_reg_v *r)   /* bwamem_pair.c:57 cal_sub */
{
	int j;
	for (j = 1; j < r->n; ++j) {
		int b_max = r->a[j].qb > r->a[0].qb ? r->a[j].qb : r->a[0].qb;
		int e_min = r->a[j].qe < r->a[0].qe ? r->a[j].qe : r->a[0].qe;
		if (e_min > b_max) {
			int min_l = r->a[j].qe - r->a[j].qb < r->a[0].qe - r->a[0].qb ? r->a[j].qe - r->a[j].qb : r->a[0].qe - r->a[0].qb;
			if (e_min - b_max >= min_l * opt->mask_level) break;
		}
	}
	return j < r->n ? r->a[j].score : opt->min_seed_len * opt->a;
}

void ora_pestat(const ora_opt_t *opt, int64_t l_pac, int n, const ora_reg_v *regs, ora_pestat_t pes[4])   /* bwamem_pair.c:72 */
{
	int i, d, max;
	struct { size_t n, m; uint64_t *a; } isize[4];
	memset(pes, 0, 4 * sizeof(ora_pestat_t));
	memset(isize, 0, sizeof isize);
	for (i = 0; i < n >> 1; ++i) {
		int dir;
		int64_t is;
		const ora_reg_v *r0 = &regs[i << 1 | 0], *r1 = &regs[i << 1 | 1];
		if (r0->n == 0 || r1->n == 0) continue;
		if (best_sub(opt, r0) > 0.8 * r0->a[0].score) continue;      /* MIN_RATIO */
		if (best_sub(opt, r1) > 0.8 * r1->a[0].score) continue;
		if (r0->a[0].rid != r1->a[0].rid) continue;
		dir = infer_dir(l_pac, r0->a[0].rb, r1->a[0].rb, &is);
		if (is && is <= opt->max_ins) {
			if (isize[dir].n == isize[dir].m) {
				isize[dir].m = isize[dir].m ? isize[dir].m << 1 : 2;
				isize[dir].a = (uint64_t*)realloc(isize[dir].a, 8 * isize[dir].m);
			}
			isize[dir].a[isize[dir].n++] = is;
		}
	}
	for (d = 0; d < 4; ++d) {
		ora_pestat_t *r = &pes[d];
		uint64_t *q = isize[d].a;
		size_t qn = isize[d].n;
		int p25, p50, p75, x;
		if (qn < 10) { r->failed = 1; free(q); isize[d].a = 0; continue; }   /* MIN_DIR_CNT */
		ora_sort_u64(qn, q);
		p25 = (int)q[(int)(.25 * qn + .499)];
		p50 = (int)q[(int)(.50 * qn + .499)];
		p75 = (int)q[(int)(.75 * qn + .499)];
		(void)p50;
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);              /* OUTLIER_BOUND */
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		for (i = x = 0, r->avg = 0; i < (int)qn; ++i)
			if (q[i] >= (uint64_t)r->low && q[i] <= (uint64_t)r->high) r->avg += q[i], ++x;
		r->avg /= x;
		for (i = 0, r->std = 0; i < (int)qn; ++i)
			if (q[i] >= (uint64_t)r->low && q[i] <= (uint64_t)r->high) r->std += (q[i] - r->avg) * (q[i] - r->avg);
		r->std = sqrt(r->std / x);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);              /* MAPPING_BOUND */
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);    /* MAX_STDDEV */
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
		free(q); isize[d].a = 0;
	}
	for (d = 0, max = 0; d < 4; ++d) max = max > (int)isize[d].n ? max : (int)isize[d].n;
	for (d = 0; d < 4; ++d)
		if (pes[d].failed == 0 && isize[d].n < max * 0.05) pes[d].failed = 1;   /* MIN_DIR_RATIO */
}

int ora_matesw(const ora_opt_t *opt, const ora_ref_t *ref, const ora_pestat_t pes[4], const ora_reg_t *a, int l_ms, const uint8_t *ms, ora_reg_v *ma)   /* bwamem_pair.c:137 */
{
	int64_t l_pac = ref->l_pac;
	int i, r, skip[4], n = 0, rid = -1;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (i = 0; i < ma->n; ++i) {
		int64_t dist;
		r = infer_dir(l_pac, a->rb, ma->a[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		int is_rev, is_larger;
		uint8_t *seq, *rev = 0, *ref_seq = 0;
		int64_t rb, re;
		if (skip[r]) continue;
		is_rev = (r >> 1 != (r & 1));
		is_larger = !(r >> 1);
		if (is_rev) {
			rev = (uint8_t*)malloc(l_ms);
			for (i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			seq = rev;
		} else seq = (uint8_t*)ms;
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) ref_seq = ora_fetch_seq(ref, &rb, (rb + re) >> 1, &re, &rid);
		if (a->rid == rid && re - rb >= opt->min_seed_len) {
			ora_kswr_t aln;
			ora_reg_t b;
			int tmp, xtra = ORA_KSW_XSUBO | ORA_KSW_XSTART | (l_ms * opt->a < 250 ? ORA_KSW_XBYTE : 0) | (opt->min_seed_len * opt->a);
			aln = ora_ksw_align2(l_ms, seq, (int)(re - rb), ref_seq, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, xtra);
			memset(&b, 0, sizeof(b));
			if (aln.score >= opt->min_seed_len && aln.qb >= 0) {
				b.rid = a->rid;
				b.is_alt = a->is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score;
				b.csub = aln.score2;
				b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				if (ma->n == ma->m) { ma->m = ma->m ? ma->m << 1 : 2; ma->a = (ora_reg_t*)realloc(ma->a, ma->m * sizeof(ora_reg_t)); }
				ma->a[ma->n++] = b;
				for (i = 0; i < ma->n - 1; ++i)
					if (ma->a[i].score < b.score) break;
				tmp = i;
				for (i = ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i-1];
				ma->a[i] = b;
			}
			++n;
		}
		if (n) ma->n = ora_sort_dedup_patch(opt, 0, 0, ma->n, ma->a);
		free(rev);
		free(ref_seq);
	}
	return n;
}

int ora_pair(const ora_opt_t *opt, const ora_ref_t *ref, const ora_pestat_t pes[4], ora_read_t s[2], ora_reg_v a[2], int id,
             int *sub, int *n_sub, int z[2], int n_pri[2])   /* bwamem_pair.c:208 */
{
	struct { size_t n, m; ora_pair64_t *a; } v = { 0, 0, 0 }, u = { 0, 0, 0 };
	int r, i, k, y[4], ret;
	int64_t l_pac = ref->l_pac;
	(void)s;
	for (r = 0; r < 2; ++r) {
		for (i = 0; i < n_pri[r]; ++i) {
			ora_pair64_t key;
			ora_reg_t *e = &a[r].a[i];
			key.x = e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb;
			key.x = (uint64_t)e->rid << 32 | (key.x - ref->anns[e->rid].offset);
			key.y = (uint64_t)e->score << 32 | i << 2 | (e->rb >= l_pac) << 1 | r;
			if (v.n == v.m) { v.m = v.m ? v.m << 1 : 2; v.a = (ora_pair64_t*)realloc(v.a, v.m * sizeof(ora_pair64_t)); }
			v.a[v.n++] = key;
		}
	}
	ora_sort_pair64(v.n, v.a);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (i = 0; i < (int)v.n; ++i) {
		for (r = 0; r < 2; ++r) {
			int dir = r << 1 | (v.a[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | ((v.a[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (k = y[which]; k >= 0; --k) {
				int64_t dist;
				int q;
				double ns;
				ora_pair64_t *p;
				if ((int)(v.a[k].y & 3) != which) continue;
				dist = (int64_t)v.a[i].x - v.a[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				ns = (dist - pes[dir].avg) / pes[dir].std;
				q = (int)((v.a[i].y >> 32) + (v.a[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt->a + .499);
				if (q < 0) q = 0;
				if (u.n == u.m) { u.m = u.m ? u.m << 1 : 2; u.a = (ora_pair64_t*)realloc(u.a, u.m * sizeof(ora_pair64_t)); }
				p = &u.a[u.n++];
				p->y = (uint64_t)k << 32 | i;
				p->x = (uint64_t)q << 32 | (ora_hash64(p->y ^ id << 8) & 0xffffffffU);   /* `id<<8` binds before `^` */
			}
		}
		y[v.a[i].y & 3] = i;
	}
	if (u.n) {
		int tmp = opt->a + opt->b;
		tmp = tmp > opt->o_del + opt->e_del ? tmp : opt->o_del + opt->e_del;
		tmp = tmp > opt->o_ins + opt->e_ins ? tmp : opt->o_ins + opt->e_ins;
		ora_sort_pair64(u.n, u.a);
		i = u.a[u.n-1].y >> 32; k = u.a[u.n-1].y << 32 >> 32;
		z[v.a[i].y & 1] = v.a[i].y << 32 >> 34;
		z[v.a[k].y & 1] = v.a[k].y << 32 >> 34;
		ret = u.a[u.n-1].x >> 32;
		*sub = u.n > 1 ? u.a[u.n-2].x >> 32 : 0;
		for (i = (long)u.n - 2, *n_sub = 0; i >= 0; --i)
			if (*sub - (int)(u.a[i].x >> 32) <= tmp) ++*n_sub;
	} else ret = 0, *sub = 0, *n_sub = 0;
	free(u.a); free(v.a);
	return ret;
}

#define RAW_MAPQ(diff, a) ((int)(6.02 * (diff) / (a) + .499))   /* bwamem_pair.c:274 */

int ora_sam_pe(const ora_opt_t *opt, const ora_ref_t *ref, const ora_pestat_t pes[4], uint64_t id, ora_read_t s[2], ora_reg_v a[2])   /* bwamem_pair.c:276 */
{
	int n = 0, i, j, z[2], o, subo, n_sub, extra_flag = 1, n_pri[2], n_aa[2];
	ora_str_t str = { 0, 0, 0 };
	ora_aln_t h[2], g[2], aa[2][2];
	memset(h, 0, sizeof h); memset(g, 0, sizeof g);
	n_aa[0] = n_aa[1] = 0;
	if (!(opt->flag & ORA_F_NO_RESCUE)) {
		ora_reg_v b[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
		for (i = 0; i < 2; ++i)
			for (j = 0; j < a[i].n; ++j)
				if (a[i].a[j].score >= a[i].a[0].score - opt->pen_unpaired) {
					if (b[i].n == b[i].m) { b[i].m = b[i].m ? b[i].m << 1 : 2; b[i].a = (ora_reg_t*)realloc(b[i].a, b[i].m * sizeof(ora_reg_t)); }
					b[i].a[b[i].n++] = a[i].a[j];
				}
		for (i = 0; i < 2; ++i)
			for (j = 0; j < b[i].n && j < opt->max_matesw; ++j)
				n += ora_matesw(opt, ref, pes, &b[i].a[j], s[!i].l_seq, (uint8_t*)s[!i].seq, &a[!i]);
		free(b[0].a); free(b[1].a);
	}
	n_pri[0] = ora_mark_primary_se(opt, a[0].n, a[0].a, id << 1 | 0);
	n_pri[1] = ora_mark_primary_se(opt, a[1].n, a[1].a, id << 1 | 1);
	if (opt->flag & ORA_F_PRIMARY5) { ora_reorder_primary5(opt->T, &a[0]); ora_reorder_primary5(opt->T, &a[1]); }
	if (opt->flag & ORA_F_NOPAIRING) goto no_pairing;
	if (n_pri[0] && n_pri[1] && (o = ora_pair(opt, ref, pes, s, a, (int)id, &subo, &n_sub, z, n_pri)) > 0) {
		int is_multi[2], q_pe, score_un, q_se[2];
		char **XA[2];
		for (i = 0; i < 2; ++i) {
			for (j = 1; j < n_pri[i]; ++j)
				if (a[i].a[j].secondary < 0 && a[i].a[j].score >= opt->T) break;
			is_multi[i] = j < n_pri[i] ? 1 : 0;
		}
		if (is_multi[0] || is_multi[1]) goto no_pairing;
		score_un = a[0].a[0].score + a[1].a[0].score - opt->pen_unpaired;
		subo = subo > score_un ? subo : score_un;
		q_pe = RAW_MAPQ(o - subo, opt->a);
		if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
		if (q_pe < 0) q_pe = 0;
		if (q_pe > 60) q_pe = 60;
		q_pe = (int)(q_pe * (1. - .5 * (a[0].a[0].frac_rep + a[1].a[0].frac_rep)) + .499);
		if (o > score_un) {
			ora_reg_t *c[2];
			c[0] = &a[0].a[z[0]]; c[1] = &a[1].a[z[1]];
			for (i = 0; i < 2; ++i) {
				if (c[i]->secondary >= 0) c[i]->sub = a[i].a[c[i]->secondary].score, c[i]->secondary = -2;
				q_se[i] = ora_approx_mapq_se(opt, c[i]);
			}
			q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
			q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
			extra_flag |= 2;
			q_se[0] = q_se[0] < RAW_MAPQ(c[0]->score - c[0]->csub, opt->a) ? q_se[0] : RAW_MAPQ(c[0]->score - c[0]->csub, opt->a);
			q_se[1] = q_se[1] < RAW_MAPQ(c[1]->score - c[1]->csub, opt->a) ? q_se[1] : RAW_MAPQ(c[1]->score - c[1]->csub, opt->a);
		} else {
			z[0] = z[1] = 0;
			q_se[0] = ora_approx_mapq_se(opt, &a[0].a[0]);
			q_se[1] = ora_approx_mapq_se(opt, &a[1].a[0]);
		}
		for (i = 0; i < 2; ++i) {
			int k = a[i].a[z[i]].secondary_all;
			if (k >= 0 && k < n_pri[i]) {
				assert(a[i].a[k].secondary_all < 0);
				for (j = 0; j < a[i].n; ++j)
					if (a[i].a[j].secondary_all == k || j == k) a[i].a[j].secondary_all = z[i];
				a[i].a[z[i]].secondary_all = -1;
			}
		}
		if (!(opt->flag & ORA_F_ALL)) {
			for (i = 0; i < 2; ++i) XA[i] = ora_gen_alt(opt, ref, &a[i], s[i].l_seq, s[i].seq);
		} else XA[0] = XA[1] = 0;
		for (i = 0; i < 2; ++i) {
			h[i] = ora_reg2aln(opt, ref, s[i].l_seq, s[i].seq, &a[i].a[z[i]]);
			h[i].mapq = (uint32_t)q_se[i] & 0xff;
			h[i].flag |= 0x40 << i | extra_flag;
			h[i].XA = XA[i] ? XA[i][z[i]] : 0;
			aa[i][n_aa[i]++] = h[i];
			if (n_pri[i] < a[i].n) {
				ora_reg_t *p = &a[i].a[n_pri[i]];
				if (p->score < opt->T || p->secondary >= 0 || !p->is_alt) continue;
				g[i] = ora_reg2aln(opt, ref, s[i].l_seq, s[i].seq, p);
				g[i].flag |= 0x800 | 0x40 << i | extra_flag;
				g[i].XA = XA[i] ? XA[i][n_pri[i]] : 0;
				aa[i][n_aa[i]++] = g[i];
			}
		}
		for (i = 0; i < n_aa[0]; ++i) ora_aln2sam(opt, ref, &str, &s[0], n_aa[0], aa[0], i, &h[1]);
		s[0].sam = strdup(str.s); str.l = 0;
		for (i = 0; i < n_aa[1]; ++i) ora_aln2sam(opt, ref, &str, &s[1], n_aa[1], aa[1], i, &h[0]);
		s[1].sam = str.s;
		if (strcmp(s[0].name, s[1].name) != 0) { fprintf(stderr, "[ora] paired reads have different names: \"%s\", \"%s\"\n", s[0].name, s[1].name); exit(1); }
		for (i = 0; i < 2; ++i) {
			free(h[i].cigar); free(g[i].cigar);
			if (XA[i] == 0) continue;
			for (j = 0; j < a[i].n; ++j) free(XA[i][j]);
			free(XA[i]);
		}
	} else goto no_pairing;
	return n;

no_pairing:
	for (i = 0; i < 2; ++i) {
		int which = -1;
		if (a[i].n) {
			if (a[i].a[0].score >= opt->T) which = 0;
			else if (n_pri[i] < a[i].n && a[i].a[n_pri[i]].score >= opt->T) which = n_pri[i];
		}
		if (which >= 0) h[i] = ora_reg2aln(opt, ref, s[i].l_seq, s[i].seq, &a[i].a[which]);
		else h[i] = ora_reg2aln(opt, ref, s[i].l_seq, s[i].seq, 0);
	}
	if (!(opt->flag & ORA_F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {
		int64_t dist;
		int d = infer_dir(ref->l_pac, a[0].a[0].rb, a[1].a[0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	ora_reg2sam(opt, ref, &s[0], &a[0], 0x41 | extra_flag, &h[1]);
	ora_reg2sam(opt, ref, &s[1], &a[1], 0x81 | extra_flag, &h[0]);
	if (strcmp(s[0].name, s[1].name) != 0) { fprintf(stderr, "[ora] paired reads have different names: \"%s\", \"%s\"\n", s[0].name, s[1].name); exit(1); }
	free(h[0].cigar); free(h[1].cigar);
	return n;
}
