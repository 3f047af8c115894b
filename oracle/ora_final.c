/* ORACLE (test infrastructure only) -- per-read finalisation and SAM text.
 * Restates bwamem.c:500-565 (mark primary), :962-986 (mapQ), :988-1010
 * (primary5 reorder), :1099-1170 (region -> CIGAR/pos), :799-956 (SAM record),
 * :1013-1059 (mem_reg2sam) and bwamem_extra.c:116-169 (XA tag).
 */
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include <assert.h>
#include "ora.h"
#include "ora_sort.h"

char ora_rg_id[256];

#define HASH_LT(a, b)  ((a).score > (b).score || ((a).score == (b).score && ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && (a).hash < (b).hash))))
ORA_SORT_DEFINE(reg_hash, ora_reg_t, HASH_LT)            /* bwamem.c:404 */
#define HASH2_LT(a, b) ((a).is_alt < (b).is_alt || ((a).is_alt == (b).is_alt && ((a).score > (b).score || ((a).score == (b).score && (a).hash < (b).hash))))
ORA_SORT_DEFINE(reg_hash2, ora_reg_t, HASH2_LT)          /* bwamem.c:407 */

typedef struct { int n, m, *a; } intv_t;
static inline void iv_push(intv_t *v, int x)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 8; v->a = (int*)realloc(v->a, sizeof(int) * v->m); }
	v->a[v->n++] = x;
}

static void mark_primary_core(const ora_opt_t *opt, int n, ora_reg_t *a, intv_t *z)   /* bwamem.c:500 */
{
	int i, k, tmp;
	tmp = opt->a + opt->b;
	tmp = opt->o_del + opt->e_del > tmp ? opt->o_del + opt->e_del : tmp;
	tmp = opt->o_ins + opt->e_ins > tmp ? opt->o_ins + opt->e_ins : tmp;
	z->n = 0;
	iv_push(z, 0);
	for (i = 1; i < n; ++i) {
		for (k = 0; k < z->n; ++k) {
			int j = z->a[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb;
			int e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * opt->mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == z->n) iv_push(z, i);
		else a[i].secondary = z->a[k];
	}
}

int ora_mark_primary_se(const ora_opt_t *opt, int n, ora_reg_t *a, int64_t id)   /* bwamem.c:528 */
{
	int i, n_pri;
	intv_t z = { 0, 0, 0 };
	if (n == 0) return 0;
	for (i = n_pri = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = ora_hash64(id + i);
		if (!a[i].is_alt) ++n_pri;
	}
	ora_isort_reg_hash(n, a);
	mark_primary_core(opt, n, a, &z);
	for (i = 0; i < n; ++i) {
		ora_reg_t *p = &a[i];
		p->secondary_all = i;
		if (!p->is_alt && p->secondary >= 0 && a[p->secondary].is_alt) p->alt_sc = a[p->secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {
		if (z.m < n) { z.m = n; z.a = (int*)realloc(z.a, sizeof(int) * n); }
		if (n_pri > 0) ora_isort_reg_hash2(n, a);
		for (i = 0; i < n; ++i) z.a[a[i].secondary_all] = i;
		for (i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) {
				a[i].secondary_all = z.a[a[i].secondary];
				if (a[i].is_alt) a[i].secondary = INT_MAX;
			} else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {
			for (i = 0; i < n_pri; ++i) a[i].sub = 0, a[i].secondary = -1;
			mark_primary_core(opt, n_pri, a, &z);
		}
	} else {
		for (i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	}
	free(z.a);
	return n_pri;
}

int ora_approx_mapq_se(const ora_opt_t *opt, const ora_reg_t *a)   /* bwamem.c:962 */
{
	int mapq, l, sub = a->sub ? a->sub : opt->min_seed_len * opt->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * opt->a - a->score) / (opt->a + opt->b) / l;
	if (a->score == 0) mapq = 0;
	else if (opt->mapQ_coef_len > 0) {
		double tmp;
		tmp = l < opt->mapQ_coef_len ? 1. : opt->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / opt->a * tmp * tmp + .499);
	} else {
		mapq = (int)(30.0 * (1. - (double)sub / a->score) * log(a->seedcov) + .499);   /* MEM_MAPQ_COEF */
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a->frac_rep) + .499);
	return mapq;
}

void ora_reorder_primary5(int T, ora_reg_v *a)   /* bwamem.c:988 */
{
	int k, n_pri = 0, left_st = INT_MAX, left_k = -1;
	ora_reg_t t;
	for (k = 0; k < a->n; ++k)
		if (a->a[k].secondary < 0 && !a->a[k].is_alt && a->a[k].score >= T) ++n_pri;
	if (n_pri <= 1) return;
	for (k = 0; k < a->n; ++k) {
		ora_reg_t *p = &a->a[k];
		if (p->secondary >= 0 || p->is_alt || p->score < T) continue;
		if (p->qb < left_st) left_st = p->qb, left_k = k;
	}
	assert(a->a[0].secondary < 0);
	if (left_k == 0) return;
	t = a->a[0]; a->a[0] = a->a[left_k]; a->a[left_k] = t;
	for (k = 1; k < a->n; ++k) {
		ora_reg_t *p = &a->a[k];
		if (p->secondary == 0) p->secondary = left_k;
		else if (p->secondary == left_k) p->secondary = 0;
		if (p->secondary_all == 0) p->secondary_all = left_k;
		else if (p->secondary_all == left_k) p->secondary_all = 0;
	}
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)   /* bwamem.c:799 */
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)(((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.));
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

ora_aln_t ora_reg2aln(const ora_opt_t *opt, const ora_ref_t *ref, int l_query, const char *query_, const ora_reg_t *ar)   /* bwamem.c:1099 */
{
	ora_aln_t a;
	int i, w2, tmp, qb, qe, NM, score, is_rev, last_sc = -(1 << 30), l_MD;
	int64_t pos, rb, re;
	uint8_t *query;
	memset(&a, 0, sizeof(a));
	if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	qb = ar->qb; qe = ar->qe; rb = ar->rb; re = ar->re;
	query = (uint8_t*)malloc(l_query);
	for (i = 0; i < l_query; ++i) query[i] = query_[i] < 5 ? query_[i] : ora_nt4_table[(uint8_t)query_[i]];
	a.mapq = ar->secondary < 0 ? (uint32_t)ora_approx_mapq_se(opt, ar) & 0xff : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_del, opt->e_del);
	w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt->a, opt->o_ins, opt->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt->w) w2 = w2 < ar->w ? w2 : ar->w;
	i = 0; a.cigar = 0;
	do {
		free(a.cigar);
		w2 = w2 < opt->w << 2 ? w2 : opt->w << 2;
		a.cigar = ora_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w2, ref->l_pac, ref->pac,
		                         qe - qb, &query[qb], rb, re, &score, &a.n_cigar, &NM);
		if (score == last_sc || w2 == opt->w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - opt->a);
	l_MD = (int)strlen((char*)(a.cigar + a.n_cigar)) + 1;
	a.NM = (uint32_t)NM & 0x3fffff;                        /* 22-bit field (bwa.h:180) */
	pos = ora_depos(ref, rb < ref->l_pac ? rb : re - 1, &is_rev);
	a.is_rev = is_rev;
	if (a.n_cigar > 0) {                                   /* squeeze a leading or trailing deletion */
		if ((a.cigar[0] & 0xf) == 2) {
			pos += a.cigar[0] >> 4;
			--a.n_cigar;
			memmove(a.cigar, a.cigar + 1, a.n_cigar * 4 + l_MD);
		} else if ((a.cigar[a.n_cigar - 1] & 0xf) == 2) {
			--a.n_cigar;
			memmove(a.cigar + a.n_cigar, a.cigar + a.n_cigar + 1, l_MD);
		}
	}
	if (qb != 0 || qe != l_query) {                        /* clipping */
		int clip5, clip3;
		clip5 = is_rev ? l_query - qe : qb;
		clip3 = is_rev ? qb : l_query - qe;
		a.cigar = (uint32_t*)realloc(a.cigar, 4 * (a.n_cigar + 2) + l_MD);
		if (clip5) {
			memmove(a.cigar + 1, a.cigar, a.n_cigar * 4 + l_MD);
			a.cigar[0] = (uint32_t)clip5 << 4 | 3;
			++a.n_cigar;
		}
		if (clip3) {
			memmove(a.cigar + a.n_cigar + 1, a.cigar + a.n_cigar, l_MD);
			a.cigar[a.n_cigar++] = (uint32_t)clip3 << 4 | 3;
		}
	}
	a.rid = ora_pos2rid(ref, pos);
	assert(a.rid == ar->rid);
	a.pos = pos - ref->anns[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	free(query);
	return a;
}

static inline int cigar_rlen(int n_cigar, const uint32_t *cigar)   /* bwamem.c:808 */
{
	int k, l;
	for (k = l = 0; k < n_cigar; ++k) {
		int op = cigar[k] & 0xf;
		if (op == 0 || op == 2) l += cigar[k] >> 4;
	}
	return l;
}

static void put_cigar(const ora_opt_t *opt, const ora_aln_t *p, ora_str_t *str, int which)   /* bwamem.c:819 */
{
	int i;
	if (p->n_cigar) {
		for (i = 0; i < p->n_cigar; ++i) {
			int c = p->cigar[i] & 0xf;
			if (!(opt->flag & ORA_F_SOFTCLIP) && !p->is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
			ora_str_putw(str, p->cigar[i] >> 4); ora_str_putc(str, "MIDSH"[c]);
		}
	} else ora_str_putc(str, '*');
}

void ora_aln2sam(const ora_opt_t *opt, const ora_ref_t *ref, ora_str_t *str, ora_read_t *s, int n, const ora_aln_t *list, int which, const ora_aln_t *m_)   /* bwamem.c:832 */
{
	int i;
	ora_aln_t ptmp = list[which], *p = &ptmp, mtmp, *m = 0;
	if (m_) mtmp = *m_, m = &mtmp;
	p->flag |= m ? 0x1 : 0;
	p->flag |= p->rid < 0 ? 0x4 : 0;
	p->flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p->rid < 0 && m && m->rid >= 0) p->rid = m->rid, p->pos = m->pos, p->is_rev = m->is_rev, p->n_cigar = 0;
	if (m && m->rid < 0 && p->rid >= 0) m->rid = p->rid, m->pos = p->pos, m->is_rev = p->is_rev, m->n_cigar = 0;
	p->flag |= p->is_rev ? 0x10 : 0;
	p->flag |= m && m->is_rev ? 0x20 : 0;
	ora_str_puts(str, s->name); ora_str_putc(str, '\t');
	ora_str_putw(str, (p->flag & 0xffff) | (p->flag & 0x10000 ? 0x100 : 0)); ora_str_putc(str, '\t');
	if (p->rid >= 0) {
		ora_str_puts(str, ref->anns[p->rid].name); ora_str_putc(str, '\t');
		ora_str_putl(str, p->pos + 1); ora_str_putc(str, '\t');
		ora_str_putw(str, p->mapq); ora_str_putc(str, '\t');
		put_cigar(opt, p, str, which);
	} else ora_str_putsn(str, "*\t0\t0\t*", 7);
	ora_str_putc(str, '\t');
	if (m && m->rid >= 0) {                                /* mate fields */
		if (p->rid == m->rid) ora_str_putc(str, '=');
		else ora_str_puts(str, ref->anns[m->rid].name);
		ora_str_putc(str, '\t');
		ora_str_putl(str, m->pos + 1); ora_str_putc(str, '\t');
		if (p->rid == m->rid) {
			int64_t p0 = p->pos + (p->is_rev ? cigar_rlen(p->n_cigar, p->cigar) - 1 : 0);
			int64_t p1 = m->pos + (m->is_rev ? cigar_rlen(m->n_cigar, m->cigar) - 1 : 0);
			if (m->n_cigar == 0 || p->n_cigar == 0) ora_str_putc(str, '0');
			else ora_str_putl(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else ora_str_putc(str, '0');
	} else ora_str_putsn(str, "*\t0\t0", 5);
	ora_str_putc(str, '\t');
	if (p->flag & 0x100) ora_str_putsn(str, "*\t*", 3);      /* SEQ / QUAL */
	else if (!p->is_rev) {
		int qb = 0, qe = s->l_seq;
		if (p->n_cigar && which && !(opt->flag & ORA_F_SOFTCLIP) && !p->is_alt) {
			if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qb += p->cigar[0] >> 4;
			if ((p->cigar[p->n_cigar-1] & 0xf) == 4 || (p->cigar[p->n_cigar-1] & 0xf) == 3) qe -= p->cigar[p->n_cigar-1] >> 4;
		}
		for (i = qb; i < qe; ++i) ora_str_putc(str, "ACGTN"[(int)s->seq[i]]);
		ora_str_putc(str, '\t');
		if (s->qual) for (i = qb; i < qe; ++i) ora_str_putc(str, s->qual[i]);
		else ora_str_putc(str, '*');
	} else {
		int qb = 0, qe = s->l_seq;
		if (p->n_cigar && which && !(opt->flag & ORA_F_SOFTCLIP) && !p->is_alt) {
			if ((p->cigar[0] & 0xf) == 4 || (p->cigar[0] & 0xf) == 3) qe -= p->cigar[0] >> 4;
			if ((p->cigar[p->n_cigar-1] & 0xf) == 4 || (p->cigar[p->n_cigar-1] & 0xf) == 3) qb += p->cigar[p->n_cigar-1] >> 4;
		}
		for (i = qe - 1; i >= qb; --i) ora_str_putc(str, "TGCAN"[(int)s->seq[i]]);
		ora_str_putc(str, '\t');
		if (s->qual) for (i = qe - 1; i >= qb; --i) ora_str_putc(str, s->qual[i]);
		else ora_str_putc(str, '*');
	}
	if (p->n_cigar) {                                      /* tags (bwamem.c:911) */
		ora_str_putsn(str, "\tNM:i:", 6); ora_str_putw(str, p->NM);
		ora_str_putsn(str, "\tMD:Z:", 6); ora_str_puts(str, (char*)(p->cigar + p->n_cigar));
	}
	if (m && m->n_cigar) { ora_str_putsn(str, "\tMC:Z:", 6); put_cigar(opt, m, str, which); }
	if (p->score >= 0) { ora_str_putsn(str, "\tAS:i:", 6); ora_str_putw(str, p->score); }
	if (p->sub >= 0) { ora_str_putsn(str, "\tXS:i:", 6); ora_str_putw(str, p->sub); }
	if (ora_rg_id[0]) { ora_str_putsn(str, "\tRG:Z:", 6); ora_str_puts(str, ora_rg_id); }
	if (!(p->flag & 0x100)) {
		for (i = 0; i < n; ++i)
			if (i != which && !(list[i].flag & 0x100)) break;
		if (i < n) {
			ora_str_putsn(str, "\tSA:Z:", 6);
			for (i = 0; i < n; ++i) {
				const ora_aln_t *r = &list[i];
				int k;
				if (i == which || (r->flag & 0x100)) continue;
				ora_str_puts(str, ref->anns[r->rid].name); ora_str_putc(str, ',');
				ora_str_putl(str, r->pos + 1); ora_str_putc(str, ',');
				ora_str_putc(str, "+-"[r->is_rev]); ora_str_putc(str, ',');
				for (k = 0; k < r->n_cigar; ++k) { ora_str_putw(str, r->cigar[k] >> 4); ora_str_putc(str, "MIDSH"[r->cigar[k] & 0xf]); }
				ora_str_putc(str, ','); ora_str_putw(str, r->mapq);
				ora_str_putc(str, ','); ora_str_putw(str, r->NM);
				ora_str_putc(str, ';');
			}
		}
		if (p->alt_sc > 0) {
			char buf[64];
			snprintf(buf, sizeof buf, "\tpa:f:%.3f", (double)p->score / p->alt_sc);
			ora_str_puts(str, buf);
		}
	}
	if (p->XA) {
		ora_str_putsn(str, (opt->flag & ORA_F_XB) ? "\tXB:Z:" : "\tXA:Z:", 6);
		ora_str_puts(str, p->XA);
	}
	if (s->comment) { ora_str_putc(str, '\t'); ora_str_puts(str, s->comment); }
	if ((opt->flag & ORA_F_REF_HDR) && p->rid >= 0 && ref->anns[p->rid].anno != 0 && ref->anns[p->rid].anno[0] != 0) {
		size_t t0;
		ora_str_putsn(str, "\tXR:Z:", 6);
		t0 = str->l;
		ora_str_puts(str, ref->anns[p->rid].anno);
		for (; t0 < str->l; ++t0) if (str->s[t0] == '\t') str->s[t0] = ' ';
	}
	ora_str_putc(str, '\n');
}

static inline int pri_idx(double XA_drop_ratio, const ora_reg_t *a, int i)   /* bwamem_extra.c:116 */
{
	int k = a[i].secondary_all;
	if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
	return -1;
}

char **ora_gen_alt(const ora_opt_t *opt, const ora_ref_t *ref, const ora_reg_v *a, int l_query, const char *query)   /* bwamem_extra.c:124 */
{
	int i, k, r, *cnt, tot;
	ora_str_t *aln = 0, str = { 0, 0, 0 };
	char **XA = 0, *has_alt;
	cnt = (int*)calloc(a->n ? a->n : 1, sizeof(int));
	has_alt = (char*)calloc(a->n ? a->n : 1, 1);
	for (i = 0, tot = 0; i < a->n; ++i) {
		r = pri_idx(opt->XA_drop_ratio, a->a, i);
		if (r >= 0) {
			++cnt[r]; ++tot;
			if (a->a[i].is_alt) has_alt[r] = 1;
		}
	}
	if (tot == 0) goto end;
	aln = (ora_str_t*)calloc(a->n, sizeof(ora_str_t));
	for (i = 0; i < a->n; ++i) {
		ora_aln_t t;
		if ((r = pri_idx(opt->XA_drop_ratio, a->a, i)) < 0) continue;
		if (cnt[r] > opt->max_XA_hits_alt || (!has_alt[r] && cnt[r] > opt->max_XA_hits)) continue;
		t = ora_reg2aln(opt, ref, l_query, query, &a->a[i]);
		str.l = 0;
		ora_str_puts(&str, ref->anns[t.rid].name);
		ora_str_putc(&str, ','); ora_str_putc(&str, "+-"[t.is_rev]); ora_str_putl(&str, t.pos + 1);
		ora_str_putc(&str, ',');
		for (k = 0; k < t.n_cigar; ++k) { ora_str_putw(&str, t.cigar[k] >> 4); ora_str_putc(&str, "MIDSHN"[t.cigar[k] & 0xf]); }
		ora_str_putc(&str, ','); ora_str_putw(&str, t.NM);
		if (opt->flag & ORA_F_XB) { ora_str_putc(&str, ','); ora_str_putw(&str, t.score); }
		ora_str_putc(&str, ';');
		free(t.cigar);
		ora_str_putsn(&aln[r], str.s, (int)str.l);
	}
	XA = (char**)calloc(a->n, sizeof(char*));
	for (k = 0; k < a->n; ++k) XA[k] = aln[k].s;
end:
	free(has_alt); free(cnt); free(aln); free(str.s);
	return XA;
}

void ora_reg2sam(const ora_opt_t *opt, const ora_ref_t *ref, ora_read_t *s, ora_reg_v *a, int extra_flag, const ora_aln_t *m)   /* bwamem.c:1013 */
{
	ora_str_t str = { 0, 0, 0 };
	struct { int n, m; ora_aln_t *a; } aa = { 0, 0, 0 };
	int k, l;
	char **XA = 0;
	if (!(opt->flag & ORA_F_ALL)) XA = ora_gen_alt(opt, ref, a, s->l_seq, s->seq);
	for (k = l = 0; k < a->n; ++k) {
		ora_reg_t *p = &a->a[k];
		ora_aln_t *q;
		if (p->score < opt->T) continue;
		if (p->secondary >= 0 && (p->is_alt || !(opt->flag & ORA_F_ALL))) continue;
		if (p->secondary >= 0 && p->secondary < INT_MAX && p->score < a->a[p->secondary].score * opt->drop_ratio) continue;
		if (aa.n == aa.m) { aa.m = aa.m ? aa.m << 1 : 2; aa.a = (ora_aln_t*)realloc(aa.a, sizeof(ora_aln_t) * aa.m); }
		q = &aa.a[aa.n++];
		*q = ora_reg2aln(opt, ref, s->l_seq, s->seq, p);
		assert(q->rid >= 0);
		q->XA = XA ? XA[k] : 0;
		q->flag |= extra_flag;
		if (p->secondary >= 0) q->sub = -1;
		if (l && p->secondary < 0) q->flag |= (opt->flag & ORA_F_NO_MULTI) ? 0x10000 : 0x800;
		if (!(opt->flag & ORA_F_KEEP_SUPP_MAPQ) && l && !p->is_alt && q->mapq > aa.a[0].mapq) q->mapq = aa.a[0].mapq;
		++l;
	}
	if (aa.n == 0) {
		ora_aln_t t = ora_reg2aln(opt, ref, s->l_seq, s->seq, 0);
		t.flag |= extra_flag;
		ora_aln2sam(opt, ref, &str, s, 1, &t, 0, m);
	} else {
		for (k = 0; k < aa.n; ++k) ora_aln2sam(opt, ref, &str, s, aa.n, aa.a, k, m);
		for (k = 0; k < aa.n; ++k) free(aa.a[k].cigar);
		free(aa.a);
	}
	s->sam = str.s;
	if (XA) { for (k = 0; k < a->n; ++k) free(XA[k]); free(XA); }
}
