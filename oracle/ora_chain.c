/* ORACLE (test infrastructure only) -- seeding and chaining.
 * Restates bwamem.c:74-110 (defaults), :137-185 (3-pass interval collection),
 * :197-322 (greedy chaining through a B-tree, kbtree.h), :334-392 (chain
 * filter), :578-622 (seed SW filter, dormant for reads < ~700 bp).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "ora.h"
#include "ora_sort.h"

void ora_fill_scmat(int a, int b, int8_t mat[25])      /* bwa.c:249 */
{
	int i, j, k = 0;
	for (i = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b;
		mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) mat[k++] = -1;
}

void ora_opt_init(ora_opt_t *o)                        /* bwamem.c:74 */
{
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100; o->T = 30; o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
	o->max_chain_gap = 10000; o->max_ins = 10000;
	o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->XA_drop_ratio = 0.80f;
	o->split_factor = 1.5f;
	o->chunk_size = 30000000;                          /* fork value (bwamem.c:99); stock bwa has 10000000 */
	o->n_threads = 1;
	o->max_XA_hits = 5; o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95f;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log(o->mapQ_coef_len);   /* int field: log(50) -> 3 */
	ora_fill_scmat(o->a, o->b, o->mat);
}

ora_aux_t *ora_aux_new(void) { return (ora_aux_t*)calloc(1, sizeof(ora_aux_t)); }
void ora_aux_free(ora_aux_t *a)
{
	free(a->mem.a); free(a->mem1.a); free(a->tmpv[0].a); free(a->tmpv[1].a); free(a);
}

#define INTV_LT(a, b) ((a).info < (b).info)            /* bwamem.c:116: key is `info` only */
ORA_SORT_DEFINE(intv, ora_intv_t, INTV_LT)

static inline void intv_push(ora_intv_v *v, const ora_intv_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 16; v->a = (ora_intv_t*)realloc(v->a, v->m * sizeof(ora_intv_t)); }
	v->a[v->n++] = *p;
}

void ora_collect_intv(const ora_opt_t *opt, const ora_fmi_t *f, int len, const uint8_t *seq, ora_aux_t *a)   /* bwamem.c:137 */
{
	int i, k, x = 0, old_n;
	int split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	ora_intv_v *tmp[2] = { &a->tmpv[0], &a->tmpv[1] };
	a->mem.n = 0;
	while (x < len) {                                   /* pass 1: all SMEMs */
		if (seq[x] < 4) {
			x = ora_smem1a(f, len, seq, x, 1, 0, &a->mem1, tmp);
			for (i = 0; i < a->mem1.n; ++i) {
				ora_intv_t *p = &a->mem1.a[i];
				int slen = (uint32_t)p->info - (p->info >> 32);
				if (slen >= opt->min_seed_len) intv_push(&a->mem, p);
			}
		} else ++x;
	}
	old_n = a->mem.n;
	for (k = 0; k < old_n; ++k) {                        /* pass 2: re-seed inside long, rare SMEMs */
		ora_intv_t p = a->mem.a[k];
		int start = p.info >> 32, end = (int32_t)p.info;
		if (end - start < split_len || p.x[2] > (uint64_t)opt->split_width) continue;
		ora_smem1a(f, len, seq, (start + end) >> 1, (int)p.x[2] + 1, 0, &a->mem1, tmp);
		for (i = 0; i < a->mem1.n; ++i)
			if ((int)((uint32_t)a->mem1.a[i].info - (a->mem1.a[i].info >> 32)) >= opt->min_seed_len)
				intv_push(&a->mem, &a->mem1.a[i]);
	}
	if (opt->max_mem_intv > 0) {                         /* pass 3: LAST-like forward seeds */
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				ora_intv_t m;
				x = ora_seed_strategy1(f, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m);
				if (m.x[2] > 0) intv_push(&a->mem, &m);
			} else ++x;
		}
	}
	ora_isort_intv(a->mem.n, a->mem.a);
}

/* ---------------------------------------------------------------- B-tree ---
 * Literal model of kbtree.h as instantiated at bwamem.c:191-194: keys are
 * chains ordered by `pos`; node capacity 2t-1 with
 * t = ((512 - 4 - 8) / (8 + sizeof(mem_chain_t)=32) + 1) >> 1 = 6  (kbtree.h:59).
 * Duplicate positions are possible, and where a duplicate lands depends on the
 * node splits, so the structure is reproduced rather than replaced by a sorted
 * array.  Nodes hold indices into a chain pool instead of the structs.
 */
#define BT_T 6
#define BT_MAXK (2 * BT_T - 1)
typedef struct bt_node {
	int is_internal, n;
	int key[BT_MAXK];
	struct bt_node *ptr[BT_MAXK + 1];
} bt_node;
typedef struct { bt_node *root; int n_keys; ora_chain_t *pool; int n_pool, m_pool; } btree;

static bt_node *bt_newnode(void) { return (bt_node*)calloc(1, sizeof(bt_node)); }
static void bt_free(bt_node *x) { int i; if (!x) return; if (x->is_internal) for (i = 0; i <= x->n; ++i) bt_free(x->ptr[i]); free(x); }

/* kbtree.h:119 __kb_getp_aux: index of the first key >= pos if it equals pos
 * (r=0), else of the greatest key < pos (r>0 when pos is greater than that
 * key or than all keys... see below), -1 if none. */
static int bt_find(const btree *b, const bt_node *x, int64_t pos, int *r)
{
	int begin = 0, end = x->n, tr;
	if (x->n == 0) return -1;
	if (!r) r = &tr;
	while (begin < end) {
		int mid = (begin + end) >> 1;
		if (b->pool[x->key[mid]].pos < pos) begin = mid + 1;
		else end = mid;
	}
	if (begin == x->n) { *r = 1; return x->n - 1; }
	*r = (b->pool[x->key[begin]].pos < pos) - (pos < b->pool[x->key[begin]].pos);   /* cmp(k, key[begin]) */
	if (*r < 0) --begin;
	return begin;
}

/* kbtree.h:152 kb_intervalp, `lower` only: greatest key <= pos (index into pool) or -1 */
static int bt_lower(const btree *b, int64_t pos)
{
	const bt_node *x = b->root;
	int lower = -1;
	while (x) {
		int r = 0, i = bt_find(b, x, pos, &r);
		if (i >= 0 && r == 0) return x->key[i];
		if (i >= 0) lower = x->key[i];
		if (!x->is_internal) return lower;
		x = x->ptr[i + 1];
	}
	return lower;
}

static void bt_split(bt_node *x, int i, bt_node *y)    /* kbtree.h:172 */
{
	bt_node *z = bt_newnode();
	z->is_internal = y->is_internal;
	z->n = BT_T - 1;
	memcpy(z->key, y->key + BT_T, sizeof(int) * (BT_T - 1));
	if (y->is_internal) memcpy(z->ptr, y->ptr + BT_T, sizeof(bt_node*) * BT_T);
	y->n = BT_T - 1;
	memmove(x->ptr + i + 2, x->ptr + i + 1, sizeof(bt_node*) * (x->n - i));
	x->ptr[i + 1] = z;
	memmove(x->key + i + 1, x->key + i, sizeof(int) * (x->n - i));
	x->key[i] = y->key[BT_T - 1];
	++x->n;
}

static void bt_put_nonfull(btree *b, bt_node *x, int k)   /* kbtree.h:188 */
{
	int64_t pos = b->pool[k].pos;
	int i;
	if (!x->is_internal) {
		i = bt_find(b, x, pos, 0);
		if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, (x->n - i - 1) * sizeof(int));
		x->key[i + 1] = k;
		++x->n;
	} else {
		i = bt_find(b, x, pos, 0) + 1;
		if (x->ptr[i]->n == BT_MAXK) {
			bt_split(x, i, x->ptr[i]);
			if (pos > b->pool[x->key[i]].pos) ++i;
		}
		bt_put_nonfull(b, x->ptr[i], k);
	}
}

static void bt_put(btree *b, int k)                     /* kbtree.h:209 */
{
	bt_node *r = b->root;
	++b->n_keys;
	if (r->n == BT_MAXK) {
		bt_node *s = bt_newnode();
		b->root = s; s->is_internal = 1; s->n = 0;
		s->ptr[0] = r;
		bt_split(s, 0, r);
		r = s;
	}
	bt_put_nonfull(b, r, k);
}

static void bt_inorder(const bt_node *x, const ora_chain_t *pool, ora_chain_v *out)   /* kbtree.h:336 */
{
	int i;
	if (!x) return;
	for (i = 0; i < x->n; ++i) {
		if (x->is_internal) bt_inorder(x->ptr[i], pool, out);
		out->a[out->n++] = pool[x->key[i]];
	}
	if (x->is_internal) bt_inorder(x->ptr[x->n], pool, out);
}

/* bwamem.c:197 test_and_merge */
static int try_merge(const ora_opt_t *opt, int64_t l_pac, ora_chain_t *c, const ora_seed_t *p, int seed_rid)
{
	int64_t qend, rend, x, y;
	const ora_seed_t *last = &c->seeds[c->n - 1];
	qend = last->qbeg + last->len;
	rend = last->rbeg + last->len;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend)
		return 1;                                       /* contained: swallow it */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0;
	x = p->qbeg - last->qbeg;
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
		if (c->n == c->m) { c->m <<= 1; c->seeds = (ora_seed_t*)realloc(c->seeds, c->m * sizeof(ora_seed_t)); }
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

int ora_chain_weight(const ora_chain_t *c)              /* bwamem.c:220 */
{
	int64_t end;
	int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const ora_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += s->qbeg + s->len - end;
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const ora_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += s->rbeg + s->len - end;
		end = end > s->rbeg + s->len ? end : s->rbeg + s->len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

ora_chain_v ora_chain(const ora_opt_t *opt, const ora_index_t *idx, int len, const uint8_t *seq, ora_aux_t *aux)   /* bwamem.c:258 */
{
	const ora_fmi_t *f = idx->fmi;
	const ora_ref_t *ref = idx->ref;
	int i, b, e, l_rep;
	int64_t l_pac = ref->l_pac;
	ora_chain_v chain = { 0, 0, 0 };
	btree tree;
	if (len < opt->min_seed_len) return chain;
	memset(&tree, 0, sizeof tree);
	tree.root = bt_newnode();
	ora_collect_intv(opt, f, len, seq, aux);
	for (i = 0, b = e = l_rep = 0; i < aux->mem.n; ++i) {      /* bwamem.c:272: union length of over-abundant seeds */
		ora_intv_t *p = &aux->mem.a[i];
		int sb = (p->info >> 32), se = (uint32_t)p->info;
		if (p->x[2] <= (uint64_t)opt->max_occ) continue;
		if (sb > e) l_rep += e - b, b = sb, e = se;
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	for (i = 0; i < aux->mem.n; ++i) {
		ora_intv_t *p = &aux->mem.a[i];
		int step, count, slen = (uint32_t)p->info - (p->info >> 32);
		int64_t k;
		step = p->x[2] > (uint64_t)opt->max_occ ? (int)(p->x[2] / opt->max_occ) : 1;
		for (k = count = 0; k < (int64_t)p->x[2] && count < opt->max_occ; k += step, ++count) {
			ora_seed_t s;
			int rid, to_add = 0, lower;
			int64_t pos;
			s.rbeg = pos = (int64_t)ora_sa(f, p->x[0] + k);
			s.qbeg = p->info >> 32;
			s.score = s.len = slen;
			rid = ora_intv2rid(ref, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			if (tree.n_keys) {
				lower = bt_lower(&tree, pos);
				if (lower < 0 || !try_merge(opt, l_pac, &tree.pool[lower], &s, rid)) to_add = 1;
			} else to_add = 1;
			if (to_add) {
				ora_chain_t *c;
				if (tree.n_pool == tree.m_pool) {
					tree.m_pool = tree.m_pool ? tree.m_pool << 1 : 16;
					tree.pool = (ora_chain_t*)realloc(tree.pool, tree.m_pool * sizeof(ora_chain_t));
				}
				c = &tree.pool[tree.n_pool];
				memset(c, 0, sizeof(*c));
				c->n = 1; c->m = 4;
				c->seeds = (ora_seed_t*)calloc(c->m, sizeof(ora_seed_t));
				c->seeds[0] = s;
				c->pos = pos;
				c->rid = rid;
				c->is_alt = !!ref->anns[rid].is_alt;
				bt_put(&tree, tree.n_pool++);
			}
		}
	}
	chain.m = tree.n_keys; chain.a = (ora_chain_t*)malloc(sizeof(ora_chain_t) * (tree.n_keys ? tree.n_keys : 1));
	bt_inorder(tree.root, tree.pool, &chain);
	for (i = 0; i < chain.n; ++i) chain.a[i].frac_rep = (float)l_rep / len;
	bt_free(tree.root); free(tree.pool);
	return chain;
}

#define CHN_BEG(ch) ((ch).seeds->qbeg)
#define CHN_END(ch) ((ch).seeds[(ch).n-1].qbeg + (ch).seeds[(ch).n-1].len)
#define FLT_LT(a, b) ((a).w > (b).w)                     /* bwamem.c:331 */
ORA_SORT_DEFINE(flt, ora_chain_t, FLT_LT)

int ora_chain_flt(const ora_opt_t *opt, int n_chn, ora_chain_t *a)   /* bwamem.c:334 */
{
	int i, k, n_kept = 0, *kept;
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		ora_chain_t *c = &a[i];
		c->first = -1; c->kept = 0;
		c->w = ora_chain_weight(c);
		if ((int)c->w < opt->min_chain_weight) free(c->seeds);
		else a[k++] = *c;
	}
	n_chn = k;
	ora_isort_flt(n_chn, a);
	kept = (int*)malloc(sizeof(int) * (n_chn ? n_chn : 1));
	a[0].kept = 3;
	kept[n_kept++] = 0;
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < n_kept; ++k) {
			int j = kept[k];
			int b_max = CHN_BEG(a[j]) > CHN_BEG(a[i]) ? CHN_BEG(a[j]) : CHN_BEG(a[i]);
			int e_min = CHN_END(a[j]) < CHN_END(a[i]) ? CHN_END(a[j]) : CHN_END(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = CHN_END(a[i]) - CHN_BEG(a[i]);
				int lj = CHN_END(a[j]) - CHN_BEG(a[j]);
				int min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {   /* int*float -> float */
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if (a[i].w < a[j].w * opt->drop_ratio && (int)(a[j].w - a[i].w) >= opt->min_seed_len << 1)
						break;
				}
			}
		}
		if (k == n_kept) {
			kept[n_kept++] = i;
			a[i].kept = large_ovlp ? 2 : 3;
		}
	}
	for (i = 0; i < n_kept; ++i) {
		ora_chain_t *c = &a[kept[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(kept);
	for (i = k = 0; i < n_chn; ++i) {                    /* bwamem.c:380: cap on kept=1/2 chains */
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < n_chn; ++i)
		if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		ora_chain_t *c = &a[i];
		if (c->kept == 0) free(c->seeds);
		else a[k++] = a[i];
	}
	return k;
}

/* bwamem.c:578 mem_seed_sw */
static int seed_sw(const ora_opt_t *opt, const ora_ref_t *r, int l_query, const uint8_t *query, const ora_seed_t *s)
{
	int qb, qe, rid;
	int64_t rb, re, mid, l_pac = r->l_pac;
	uint8_t *rseq, *qcopy;
	ora_kswr_t x;
	if (s->len >= 200) return -1;                         /* MEM_SHORT_LEN */
	qb = s->qbeg; qe = s->qbeg + s->len;
	rb = s->rbeg; re = s->rbeg + s->len;
	mid = (rb + re) >> 1;
	qb -= 50; qb = qb > 0 ? qb : 0;                       /* MEM_SHORT_EXT */
	qe += 50; qe = qe < l_query ? qe : l_query;
	rb -= 50; rb = rb > 0 ? rb : 0;
	re += 50; re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) {
		if (mid < l_pac) re = l_pac;
		else rb = l_pac;
	}
	if (qe - qb >= 200 || re - rb >= 200) return -1;
	rseq = ora_fetch_seq(r, &rb, mid, &re, &rid);
	qcopy = (uint8_t*)malloc(qe - qb);
	memcpy(qcopy, query + qb, qe - qb);
	x = ora_ksw_align2(qe - qb, qcopy, (int)(re - rb), rseq, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, ORA_KSW_XSTART);
	free(rseq); free(qcopy);
	return x.score;
}

void ora_flt_chained_seeds(const ora_opt_t *opt, const ora_ref_t *r, int l_query, const uint8_t *query, int n_chn, ora_chain_t *a)   /* bwamem.c:605 */
{
	double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_query);
	int i, j, k, min_HSP_score = (int)(opt->a * min_l + .499);
	if (min_l > 0.05f * l_query) return;
	for (i = 0; i < n_chn; ++i) {
		ora_chain_t *c = &a[i];
		for (j = k = 0; j < c->n; ++j) {
			ora_seed_t *s = &c->seeds[j];
			s->score = seed_sw(opt, r, l_query, query, s);
			if (s->score < 0 || s->score >= min_HSP_score) {
				s->score = s->score < 0 ? s->len * opt->a : s->score;
				c->seeds[k++] = *s;
			}
		}
		c->n = k;
	}
}
