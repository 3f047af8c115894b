/* ORACLE (test infrastructure only) -- chain -> alignment regions.
 * Restates bwamem.c:628-793 (mem_chain2aln), :413-496 (patch / sort / dedup),
 * :1061-1097 (mem_align1_core) and bwa.c:261-347 (bwa_gen_cigar2 incl. NM/MD).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "ora.h"
#include "ora_sort.h"

#define U64_LT(a, b) ((a) < (b))
ORA_SORT_DEFINE(u64, uint64_t, U64_LT)
void ora_sort_u64(size_t n, uint64_t *a) { ora_isort_u64(n, a); }                     /* utils.c:47 */
#define P64_LT(a, b) ((a).x < (b).x || ((a).x == (b).x && (a).y < (b).y))
ORA_SORT_DEFINE(p64, ora_pair64_t, P64_LT)
void ora_sort_pair64(size_t n, ora_pair64_t *a) { ora_isort_p64(n, a); }              /* utils.c:46 */

static inline int max_gap_for(const ora_opt_t *opt, int qlen)                         /* bwamem.c:628 cal_max_gap */
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

static ora_reg_t *reg_push(ora_reg_v *v)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 4; v->a = (ora_reg_t*)realloc(v->a, v->m * sizeof(ora_reg_t)); }
	return &v->a[v->n++];
}

void ora_chain2aln(const ora_opt_t *opt, const ora_ref_t *ref, int l_query, const uint8_t *query, const ora_chain_t *c, ora_reg_v *av)
{
	int i, k, rid, max_off[2], aw[2];
	int64_t l_pac = ref->l_pac, rmax[2], tmp, max = 0;
	const ora_seed_t *s;
	uint8_t *rseq;
	uint64_t *srt;
	if (c->n == 0) return;
	rmax[0] = l_pac << 1; rmax[1] = 0;
	for (i = 0; i < c->n; ++i) {                             /* widest window any seed could reach (bwamem.c:650) */
		const ora_seed_t *t = &c->seeds[i];
		int64_t b = t->rbeg - (t->qbeg + max_gap_for(opt, t->qbeg));
		int64_t e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + max_gap_for(opt, l_query - t->qbeg - t->len));
		rmax[0] = rmax[0] < b ? rmax[0] : b;
		rmax[1] = rmax[1] > e ? rmax[1] : e;
		if (t->len > max) max = t->len;
	}
	rmax[0] = rmax[0] > 0 ? rmax[0] : 0;
	rmax[1] = rmax[1] < l_pac << 1 ? rmax[1] : l_pac << 1;
	if (rmax[0] < l_pac && l_pac < rmax[1]) {
		if (c->seeds[0].rbeg < l_pac) rmax[1] = l_pac;
		else rmax[0] = l_pac;
	}
	rseq = ora_fetch_seq(ref, &rmax[0], c->seeds[0].rbeg, &rmax[1], &rid);
	assert(c->rid == rid);
	srt = (uint64_t*)malloc(c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | i;
	ora_sort_u64(c->n, srt);
	for (k = c->n - 1; k >= 0; --k) {                        /* best seed first */
		ora_reg_t *a;
		s = &c->seeds[(uint32_t)srt[k]];
		for (i = 0; i < av->n; ++i) {                        /* already covered by an earlier extension? (bwamem.c:678) */
			ora_reg_t *p = &av->a[i];
			int64_t rd;
			int qd, w, max_gap;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			max_gap = max_gap_for(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			max_gap = max_gap_for(opt, qd < rd ? qd : (int)rd);
			w = max_gap < p->w ? max_gap : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (i < av->n) {                                     /* bwamem.c:696 */
			for (i = k + 1; i < c->n; ++i) {
				const ora_seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) { srt[k] = 0; continue; }
		}
		a = reg_push(av);
		memset(a, 0, sizeof(*a));
		a->w = aw[0] = aw[1] = opt->w;
		a->score = a->truesc = -1;
		a->rid = c->rid;
		if (s->qbeg) {                                       /* left extension on reversed sequences (bwamem.c:722) */
			uint8_t *rs, *qs;
			int qle, tle, gtle, gscore;
			qs = (uint8_t*)malloc(s->qbeg);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			tmp = s->rbeg - rmax[0];
			rs = (uint8_t*)malloc(tmp > 0 ? tmp : 1);
			for (i = 0; i < tmp; ++i) rs[i] = rseq[tmp - 1 - i];
			for (i = 0; i < 2; ++i) {                        /* MAX_BAND_TRY */
				int prev = a->score;
				aw[0] = opt->w << i;
				a->score = ora_ksw_extend2(s->qbeg, qs, (int)tmp, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins,
				                           aw[0], opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &max_off[0]);
				if (a->score == prev || max_off[0] < (aw[0] >> 1) + (aw[0] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) {
				a->qb = s->qbeg - qle; a->rb = s->rbeg - tle;
				a->truesc = a->score;
			} else {
				a->qb = 0; a->rb = s->rbeg - gtle;
				a->truesc = gscore;
			}
			free(qs); free(rs);
		} else a->score = a->truesc = s->len * opt->a, a->qb = 0, a->rb = s->rbeg;
		if (s->qbeg + s->len != l_query) {                   /* right extension (bwamem.c:753) */
			int qle, tle, qe, re, gtle, gscore, sc0 = a->score;
			qe = s->qbeg + s->len;
			re = (int)(s->rbeg + s->len - rmax[0]);
			assert(re >= 0);
			for (i = 0; i < 2; ++i) {
				int prev = a->score;
				aw[1] = opt->w << i;
				a->score = ora_ksw_extend2(l_query - qe, query + qe, (int)(rmax[1] - rmax[0] - re), rseq + re, 5, opt->mat,
				                           opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw[1], opt->pen_clip3, opt->zdrop, sc0,
				                           &qle, &tle, &gtle, &gscore, &max_off[1]);
				if (a->score == prev || max_off[1] < (aw[1] >> 1) + (aw[1] >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) {
				a->qe = qe + qle; a->re = rmax[0] + re + tle;
				a->truesc += a->score - sc0;
			} else {
				a->qe = l_query; a->re = rmax[0] + re + gtle;
				a->truesc += gscore - sc0;
			}
		} else a->qe = l_query, a->re = s->rbeg + s->len;
		for (i = 0, a->seedcov = 0; i < c->n; ++i) {         /* bwamem.c:782 */
			const ora_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
				a->seedcov += t->len;
		}
		a->w = aw[0] > aw[1] ? aw[0] : aw[1];
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* ---- bwa.c:261 bwa_gen_cigar2 ---- */
static void str_need(ora_str_t *s, size_t extra)
{
	if (s->l + extra + 1 > s->m) {
		s->m = s->l + extra + 2;
		s->m--; s->m |= s->m >> 1; s->m |= s->m >> 2; s->m |= s->m >> 4; s->m |= s->m >> 8; s->m |= s->m >> 16; s->m++;
		s->s = (char*)realloc(s->s, s->m);
	}
}
void ora_str_putc(ora_str_t *s, int c) { str_need(s, 1); s->s[s->l++] = c; s->s[s->l] = 0; }
void ora_str_putsn(ora_str_t *s, const char *p, int l) { str_need(s, l); memcpy(s->s + s->l, p, l); s->l += l; s->s[s->l] = 0; }
void ora_str_puts(ora_str_t *s, const char *p) { ora_str_putsn(s, p, (int)strlen(p)); }
void ora_str_putw(ora_str_t *s, int v)
{
	char buf[16];
	int l = 0;
	unsigned x = v < 0 ? -(unsigned)v : (unsigned)v;
	if (v == 0) { ora_str_putc(s, '0'); return; }
	for (; x; x /= 10) buf[l++] = x % 10 + '0';
	if (v < 0) buf[l++] = '-';
	str_need(s, l);
	while (l > 0) s->s[s->l++] = buf[--l];
	s->s[s->l] = 0;
}
void ora_str_putl(ora_str_t *s, long v)
{
	char buf[32];
	int l = 0;
	unsigned long x = v < 0 ? -(unsigned long)v : (unsigned long)v;
	if (v == 0) { ora_str_putc(s, '0'); return; }
	for (; x; x /= 10) buf[l++] = x % 10 + '0';
	if (v < 0) buf[l++] = '-';
	str_need(s, l);
	while (l > 0) s->s[s->l++] = buf[--l];
	s->s[s->l] = 0;
}

uint32_t *ora_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                         int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, int *n_cigar, int *NM)
{
	uint32_t *cigar = 0;
	uint8_t tmp, *rseq;
	int i;
	int64_t rlen;
	if (n_cigar) *n_cigar = 0;
	if (NM) *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	rseq = ora_get_seq(l_pac, pac, rb, re, &rlen);
	if (re - rb != rlen) goto done;
	if (rb >= l_pac) {                                       /* reverse both so indels are left-aligned on the forward strand */
		for (i = 0; i < l_query >> 1; ++i) tmp = query[i], query[i] = query[l_query-1-i], query[l_query-1-i] = tmp;
		for (i = 0; i < rlen >> 1; ++i) tmp = rseq[i], rseq[i] = rseq[rlen-1-i], rseq[rlen-1-i] = tmp;
	}
	if (l_query == re - rb && w_ == 0) {                     /* no gap: no DP (bwa.c:281) */
		if (n_cigar) {
			cigar = (uint32_t*)malloc(4);
			cigar[0] = (uint32_t)l_query << 4 | 0;
			*n_cigar = 1;
		}
		for (i = 0, *score = 0; i < l_query; ++i) *score += mat[rseq[i] * 5 + query[i]];
	} else {
		int w, max_gap, max_ins, max_del, min_w;
		max_ins = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
		max_del = (int)((double)(((l_query + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
		max_gap = max_ins > max_del ? max_ins : max_del;
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		*score = ora_ksw_global2(l_query, query, (int)rlen, rseq, 5, mat, o_del, e_del, o_ins, e_ins, w, n_cigar, &cigar);
	}
	if (NM && n_cigar) {                                     /* NM and MD, MD appended after the CIGAR words (bwa.c:309) */
		int k, x, y, u, n_mm = 0, n_gap = 0;
		ora_str_t str;
		const char *int2base = rb < l_pac ? "ACGTN" : "TGCAN";
		str.l = str.m = *n_cigar * 4; str.s = (char*)cigar;
		for (k = 0, x = y = u = 0; k < *n_cigar; ++k) {
			int op, len;
			cigar = (uint32_t*)str.s;
			op = cigar[k] & 0xf; len = cigar[k] >> 4;
			if (op == 0) {
				for (i = 0; i < len; ++i) {
					if (query[x + i] != rseq[y + i]) {
						ora_str_putw(&str, u);
						ora_str_putc(&str, int2base[rseq[y + i]]);
						++n_mm; u = 0;
					} else ++u;
				}
				x += len; y += len;
			} else if (op == 2) {
				if (k > 0 && k < *n_cigar - 1) {
					ora_str_putw(&str, u); ora_str_putc(&str, '^');
					for (i = 0; i < len; ++i) ora_str_putc(&str, int2base[rseq[y + i]]);
					u = 0; n_gap += len;
				}
				y += len;
			} else if (op == 1) x += len, n_gap += len;
		}
		ora_str_putw(&str, u); ora_str_putc(&str, 0);
		*NM = n_mm + n_gap;
		cigar = (uint32_t*)str.s;
	}
	if (rb >= l_pac)
		for (i = 0; i < l_query >> 1; ++i) tmp = query[i], query[i] = query[l_query-1-i], query[l_query-1-i] = tmp;
done:
	free(rseq);
	return cigar;
}

/* ---- bwamem.c:398-402 sort keys ---- */
#define REG_END_LT(a, b) ((a).re < (b).re)
ORA_SORT_DEFINE(reg_end, ora_reg_t, REG_END_LT)
#define REG_SC_LT(a, b) ((a).score > (b).score || ((a).score == (b).score && ((a).rb < (b).rb || ((a).rb == (b).rb && (a).qb < (b).qb))))
ORA_SORT_DEFINE(reg_sc, ora_reg_t, REG_SC_LT)

static int patch_reg(const ora_opt_t *opt, const ora_ref_t *ref, uint8_t *query, const ora_reg_t *a, const ora_reg_t *b, int *w_)   /* bwamem.c:413 */
{
	int w, score, q_s, r_s;
	double r;
	if (ref == 0 || query == 0) return 0;
	assert(a->rid == b->rid && a->rb <= b->rb);
	if (a->rb < ref->l_pac && b->rb >= ref->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb));
	w = w > 0 ? w : -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb);
	r = r > 0. ? r : -r;
	if (a->re < b->rb || a->qe < b->qb) {
		if (w > opt->w << 1 || r >= 0.05f) return 0;             /* PATCH_MAX_R_BW */
	} else if (w > opt->w << 2 || r >= 0.05f * 2) return 0;
	w += a->w + b->w;
	w = w < opt->w << 2 ? w : opt->w << 2;
	ora_gen_cigar2(opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, ref->l_pac, ref->pac,
	               b->qe - a->qb, query + a->qb, a->rb, b->re, &score, 0, 0);
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;   /* PATCH_MIN_SC_RATIO */
	*w_ = w;
	return score;
}

int ora_sort_dedup_patch(const ora_opt_t *opt, const ora_ref_t *ref, uint8_t *query, int n, ora_reg_t *a)   /* bwamem.c:444 */
{
	int m, i, j;
	if (n <= 1) return n;
	ora_isort_reg_end(n, a);
	for (i = 0; i < n; ++i) a[i].n_comp = 1;
	for (i = 1; i < n; ++i) {
		ora_reg_t *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + opt->max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			ora_reg_t *q = &a[j];
			int64_t orr, oq, mr, mq;
			int score, w;
			if (q->qe == q->qb) continue;
			orr = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (orr > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, ref, query, q, p, &w)) > 0) {
				p->n_comp += q->n_comp + 1;
				p->seedcov = p->seedcov > q->seedcov ? p->seedcov : q->seedcov;
				p->sub = p->sub > q->sub ? p->sub : q->sub;
				p->csub = p->csub > q->csub ? p->csub : q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	n = m;
	ora_isort_reg_sc(n, a);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i-1].score && a[i].rb == a[i-1].rb && a[i].qb == a[i-1].qb) a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) { if (m != i) a[m++] = a[i]; else ++m; }
	return m;
}

ora_reg_v ora_align1_core(const ora_opt_t *opt, const ora_index_t *idx, int l_seq, char *seq, ora_aux_t *aux)   /* bwamem.c:1061 */
{
	int i;
	ora_chain_v chn;
	ora_reg_v regs = { 0, 0, 0 };
	for (i = 0; i < l_seq; ++i) seq[i] = seq[i] < 4 ? seq[i] : ora_nt4_table[(uint8_t)seq[i]];
	chn = ora_chain(opt, idx, l_seq, (uint8_t*)seq, aux);
	chn.n = ora_chain_flt(opt, chn.n, chn.a);
	ora_flt_chained_seeds(opt, idx->ref, l_seq, (uint8_t*)seq, chn.n, chn.a);
	for (i = 0; i < chn.n; ++i) {
		ora_chain2aln(opt, idx->ref, l_seq, (uint8_t*)seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = ora_sort_dedup_patch(opt, idx->ref, (uint8_t*)seq, regs.n, regs.a);
	for (i = 0; i < regs.n; ++i) {
		ora_reg_t *p = &regs.a[i];
		if (p->rid >= 0 && idx->ref->anns[p->rid].is_alt) p->is_alt = 1;
	}
	return regs;
}
