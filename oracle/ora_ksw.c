/* ORACLE (test infrastructure only) -- Smith-Waterman kernels.
 * Restates ksw.c of the reference: ksw_extend2 (ksw.c:380-479), ksw_global2
 * (ksw.c:504-606) and the SSE2 striped local alignment ksw_align2 with its
 * u8 / i16 workers (ksw.c:64-110 profile, :111-231 u8, :232-335 i16, :343-365).
 * The striped code is emulated lane by lane in scalar C so that its
 * result-visible quirks (lazy-F loop, saturation, tie rules) are reproduced.
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "ora.h"

typedef struct { int32_t h, e; } cell_t;

int ora_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *qle_, int *tle_, int *gtle_, int *gscore_, int *max_off_)
{
	cell_t *row;
	int8_t *prof;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int beg, end, best, best_i, best_j, max_ins, max_del, best_ie, gscore, max_off;
	assert(h0 > 0);
	prof = (int8_t*)malloc((size_t)qlen * m);
	row = (cell_t*)calloc(qlen + 1, sizeof(cell_t));
	for (k = i = 0; k < m; ++k) {                       /* query profile (ksw.c:391) */
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) prof[i++] = p[query[j]];
	}
	row[0].h = h0; row[1].h = h0 > oe_ins ? h0 - oe_ins : 0;   /* first row (ksw.c:396) */
	for (j = 2; j <= qlen && row[j-1].h > e_ins; ++j) row[j].h = row[j-1].h - e_ins;
	for (i = 0, best = 0; i < m * m; ++i) best = best > mat[i] ? best : mat[i];   /* clamp the band (ksw.c:399-407) */
	max_ins = (int)((double)(qlen * best + end_bonus - o_ins) / e_ins + 1.);
	max_ins = max_ins > 1 ? max_ins : 1;
	w = w < max_ins ? w : max_ins;
	max_del = (int)((double)(qlen * best + end_bonus - o_del) / e_del + 1.);
	max_del = max_del > 1 ? max_del : 1;
	w = w < max_del ? w : max_del;
	best = h0; best_i = best_j = -1; best_ie = -1; gscore = -1; max_off = 0;
	beg = 0; end = qlen;
	for (i = 0; i < tlen; ++i) {
		int t, f = 0, h1, rowmax = 0, rowmax_j = -1;
		const int8_t *s = &prof[target[i] * qlen];
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
		else h1 = 0;
		for (j = beg; j < end; ++j) {
			cell_t *p = &row[j];
			int h, M = p->h, e = p->e;
			p->h = h1;
			M = M ? M + s[j] : 0;                       /* ksw.c:433 */
			h = M > e ? M : e;
			h = h > f ? h : f;
			h1 = h;
			rowmax_j = rowmax > h ? rowmax_j : j;       /* last column wins ties (ksw.c:437) */
			rowmax = rowmax > h ? rowmax : h;
			t = M - oe_del; t = t > 0 ? t : 0;
			e -= e_del; e = e > t ? e : t;
			p->e = e;
			t = M - oe_ins; t = t > 0 ? t : 0;
			f -= e_ins; f = f > t ? f : t;
		}
		row[end].h = h1; row[end].e = 0;
		if (j == qlen) {                                 /* ksw.c:450: later row wins ties */
			best_ie = gscore > h1 ? best_ie : i;
			gscore = gscore > h1 ? gscore : h1;
		}
		if (rowmax == 0) break;
		if (rowmax > best) {
			best = rowmax; best_i = i; best_j = rowmax_j;
			max_off = max_off > abs(rowmax_j - i) ? max_off : abs(rowmax_j - i);
		} else if (zdrop > 0) {
			if (i - best_i > rowmax_j - best_j) {
				if (best - rowmax - ((i - best_i) - (rowmax_j - best_j)) * e_del > zdrop) break;
			} else {
				if (best - rowmax - ((rowmax_j - best_j) - (i - best_i)) * e_ins > zdrop) break;
			}
		}
		for (j = beg; j < end && row[j].h == 0 && row[j].e == 0; ++j);   /* ksw.c:466 */
		beg = j;
		for (j = end; j >= beg && row[j].h == 0 && row[j].e == 0; --j);
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(row); free(prof);
	if (qle_) *qle_ = best_j + 1;
	if (tle_) *tle_ = best_i + 1;
	if (gtle_) *gtle_ = best_ie + 1;
	if (gscore_) *gscore_ = gscore;
	if (max_off_) *max_off_ = max_off;
	return best;
}

#define NEG_INF (-0x40000000)

static uint32_t *cigar_push(int *n, int *m, uint32_t *cg, int op, int len)   /* ksw.c:491 */
{
	if (*n == 0 || op != (int)(cg[*n - 1] & 0xf)) {
		if (*n == *m) { *m = *m ? *m << 1 : 4; cg = (uint32_t*)realloc(cg, (size_t)*m << 2); }
		cg[(*n)++] = (uint32_t)len << 4 | op;
	} else cg[*n - 1] += (uint32_t)len << 4;
	return cg;
}

int ora_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
	cell_t *row;
	int8_t *prof;
	int i, j, k, oe_del = o_del + e_del, oe_ins = o_ins + e_ins, score, n_col;
	uint8_t *z;
	int want = n_cigar_ && cigar_;
	if (n_cigar_) *n_cigar_ = 0;
	n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	z = want ? (uint8_t*)malloc((size_t)n_col * tlen + 1) : 0;
	prof = (int8_t*)malloc((size_t)qlen * m);
	row = (cell_t*)calloc(qlen + 1, sizeof(cell_t));
	for (k = i = 0; k < m; ++k) {
		const int8_t *p = &mat[k * m];
		for (j = 0; j < qlen; ++j) prof[i++] = p[query[j]];
	}
	row[0].h = 0; row[0].e = NEG_INF;
	for (j = 1; j <= qlen && j <= w; ++j) row[j].h = -(o_ins + e_ins * j), row[j].e = NEG_INF;
	for (; j <= qlen; ++j) row[j].h = row[j].e = NEG_INF;
	for (i = 0; i < tlen; ++i) {
		int32_t f = NEG_INF, h1, beg, end, t;
		const int8_t *s = &prof[target[i] * qlen];
		uint8_t *zi = want ? &z[(size_t)i * n_col] : 0;
		beg = i > w ? i - w : 0;
		end = i + w + 1 < qlen ? i + w + 1 : qlen;
		h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		for (j = beg; j < end; ++j) {
			cell_t *p = &row[j];
			int32_t h, M = p->h, e = p->e;
			uint8_t d;
			p->h = h1;
			M += s[j];
			d = M >= e ? 0 : 1;                           /* ties prefer M, then E over F (ksw.c:551-554) */
			h = M >= e ? M : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			h1 = h;
			t = M - oe_del;
			e -= e_del;
			d |= e > t ? 1 << 2 : 0;
			e = e > t ? e : t;
			p->e = e;
			t = M - oe_ins;
			f -= e_ins;
			d |= f > t ? 2 << 4 : 0;
			f = f > t ? f : t;
			if (zi) zi[j - beg] = d;
		}
		row[end].h = h1; row[end].e = NEG_INF;
	}
	score = row[qlen].h;
	if (want) {                                           /* backtrack (ksw.c:586-603) */
		int n_cigar = 0, m_cigar = 0, which = 0;
		uint32_t *cigar = 0, tmp;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) cigar = cigar_push(&n_cigar, &m_cigar, cigar, 0, 1), --i, --k;
			else if (which == 1) cigar = cigar_push(&n_cigar, &m_cigar, cigar, 2, 1), --i;
			else cigar = cigar_push(&n_cigar, &m_cigar, cigar, 1, 1), --k;
		}
		if (i >= 0) cigar = cigar_push(&n_cigar, &m_cigar, cigar, 2, i + 1);
		if (k >= 0) cigar = cigar_push(&n_cigar, &m_cigar, cigar, 1, k + 1);
		for (i = 0; i < n_cigar >> 1; ++i) tmp = cigar[i], cigar[i] = cigar[n_cigar-1-i], cigar[n_cigar-1-i] = tmp;
		*n_cigar_ = n_cigar; *cigar_ = cigar;
	}
	free(row); free(prof); free(z);
	return score;
}

/* ------------------------------------------------------------------------
 * Striped local alignment, emulated.  A "vector" is P lanes (16 x u8 or
 * 8 x i16); query position of (segment j, lane l) is j + l*slen (ksw.c:87-89).
 * ---------------------------------------------------------------------- */
typedef struct {
	int qlen, slen, size, P;
	int shift, mdiff, max;          /* ksw.c:80-86 (u8 bias) */
	int *prof;                      /* [m][slen][P] */
	int *H0, *H1, *E, *Hmax;        /* [slen][P] */
} sprof_t;

static sprof_t *sprof_new(int size, int qlen, const uint8_t *query, int m, const int8_t *mat)
{
	sprof_t *q = (sprof_t*)calloc(1, sizeof(*q));
	int a, i, k, l, mn = 127, mx = 0;
	size = size > 1 ? 2 : 1;
	q->size = size; q->P = 8 * (3 - size); q->qlen = qlen;
	q->slen = (qlen + q->P - 1) / q->P;
	for (a = 0; a < m * m; ++a) {                         /* ksw.c:80-86, evaluated in int8/uint8 as there */
		if (mat[a] < (int8_t)mn) mn = (uint8_t)mat[a];
		if (mat[a] > (int8_t)mx) mx = (uint8_t)mat[a];
	}
	q->max = mx;
	q->shift = (256 - mn) & 0xff;
	q->mdiff = (mx + q->shift) & 0xff;
	q->prof = (int*)malloc(sizeof(int) * m * q->slen * q->P);
	q->H0 = (int*)calloc(4 * q->slen * q->P, sizeof(int));
	q->H1 = q->H0 + q->slen * q->P; q->E = q->H1 + q->slen * q->P; q->Hmax = q->E + q->slen * q->P;
	for (a = 0; a < m; ++a)
		for (i = 0; i < q->slen; ++i)
			for (l = 0, k = i; l < q->P; ++l, k += q->slen) {
				int v = k >= qlen ? 0 : mat[a * m + query[k]];
				q->prof[(a * q->slen + i) * q->P + l] = size == 1 ? ((v + q->shift) & 0xff) : v;
			}
	return q;
}
static void sprof_free(sprof_t *q) { free(q->prof); free(q->H0); free(q); }

static inline int sat_add_u8(int a, int b) { int s = a + b; return s > 255 ? 255 : s; }
static inline int sat_sub_u(int a, int b) { int s = a - b; return s < 0 ? 0 : s; }
static inline int sat_add_i16(int a, int b) { int s = a + b; return s > 32767 ? 32767 : s < -32768 ? -32768 : s; }

/* ksw.c:111 (u8) and ksw.c:232 (i16) share this body; `is8` picks the arithmetic. */
static ora_kswr_t striped_sw(sprof_t *q, int tlen, const uint8_t *target, int o_del, int e_del_, int o_ins, int e_ins_, int xtra)
{
	const int P = q->P, slen = q->slen, is8 = q->size == 1;
	int i, j, l, k, te = -1, gmax = 0, minsc, endsc, n_b = 0, m_b = 0;
	int oe_del = o_del + e_del_, oe_ins = o_ins + e_ins_, e_del = e_del_, e_ins = e_ins_;
	uint64_t *b = 0;
	int *H0 = q->H0, *H1 = q->H1, *E = q->E, *Hmax = q->Hmax, *sw;
	int *h = (int*)malloc(sizeof(int) * P * 4), *e = h + P, *f = e + P, *mx = f + P;
	ora_kswr_t r = { 0, -1, -1, -1, -1, -1, -1 };                 /* ksw.c:44 g_defr */
	minsc = (xtra & ORA_KSW_XSUBO) ? xtra & 0xffff : 0x10000;
	endsc = (xtra & ORA_KSW_XSTOP) ? xtra & 0xffff : 0x10000;
	if (is8) { oe_del &= 0xff; oe_ins &= 0xff; e_del &= 0xff; e_ins &= 0xff; }
	memset(E, 0, sizeof(int) * slen * P); memset(H0, 0, sizeof(int) * slen * P); memset(Hmax, 0, sizeof(int) * slen * P);
	for (i = 0; i < tlen; ++i) {
		const int *S = q->prof + target[i] * slen * P;
		int imax, done;
		for (l = 0; l < P; ++l) f[l] = 0, mx[l] = 0;
		h[0] = 0;                                                 /* h = H0[slen-1] shifted by one lane */
		for (l = 1; l < P; ++l) h[l] = H0[(slen - 1) * P + l - 1];
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < P; ++l) {
				int hv, ev = E[j * P + l], t;
				if (is8) { hv = sat_add_u8(h[l], S[j * P + l]); hv = sat_sub_u(hv, q->shift); }
				else hv = sat_add_i16(h[l], S[j * P + l]);
				hv = hv > ev ? hv : ev;
				hv = hv > f[l] ? hv : f[l];
				mx[l] = mx[l] > hv ? mx[l] : hv;
				H1[j * P + l] = hv;
				ev = sat_sub_u(ev, e_del); t = sat_sub_u(hv, oe_del);
				E[j * P + l] = ev > t ? ev : t;
				f[l] = sat_sub_u(f[l], e_ins); t = sat_sub_u(hv, oe_ins);
				f[l] = f[l] > t ? f[l] : t;
				h[l] = H0[j * P + l];
			}
		}
		for (k = 0, done = 0; k < 16 && !done; ++k) {             /* lazy-F loop (ksw.c:176-188 / 276-286) */
			for (l = P - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int any = 0;
				for (l = 0; l < P; ++l) {
					int hv = H1[j * P + l];
					hv = hv > f[l] ? hv : f[l];
					H1[j * P + l] = hv;
					hv = sat_sub_u(hv, oe_ins);
					f[l] = sat_sub_u(f[l], e_ins);
					if (f[l] > hv) any = 1;                       /* u8: subs(f,h)!=0 ; i16: f>h */
				}
				if (!any) { done = 1; break; }
			}
		}
		for (l = 0, imax = 0; l < P; ++l) imax = imax > mx[l] ? imax : mx[l];
		if (imax >= minsc) {
			if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
				if (n_b == m_b) { m_b = m_b ? m_b << 1 : 8; b = (uint64_t*)realloc(b, 8 * m_b); }
				b[n_b++] = (uint64_t)imax << 32 | i;
			} else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | i;
		}
		if (imax > gmax) {
			gmax = imax; te = i;
			memcpy(Hmax, H1, sizeof(int) * slen * P);
			if (is8 ? (gmax + q->shift >= 255 || gmax >= endsc) : (gmax >= endsc)) break;
		}
		sw = H1; H1 = H0; H0 = sw;
	}
	r.score = is8 ? (gmax + q->shift < 255 ? gmax : 255) : gmax;
	r.te = te;
	if (!is8 || r.score != 255) {
		int max = -1, tmp, low, high, n = slen * P;
		if (!is8) r.qe = -1;
		for (i = 0; i < n; ++i) {                                  /* memory order: segment-major, lane-minor */
			int v = Hmax[i];
			if (v > max) max = v, r.qe = i / P + i % P * slen;
			else if (v == max && (tmp = i / P + i % P * slen) < r.qe) r.qe = tmp;
		}
		if (b) {
			i = (r.score + q->max - 1) / q->max;
			low = te - i; high = te + i;
			for (i = 0; i < n_b; ++i) {
				int e2 = (int32_t)b[i];
				if ((e2 < low || e2 > high) && (int)(b[i] >> 32) > r.score2) r.score2 = b[i] >> 32, r.te2 = e2;
			}
		}
	}
	free(b); free(h);
	return r;
}

static void rev_bytes(int l, uint8_t *s) { int i; for (i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l-1-i]; s[l-1-i] = t; } }

ora_kswr_t ora_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	sprof_t *q = sprof_new((xtra & ORA_KSW_XBYTE) ? 1 : 2, qlen, query, m, mat);
	int size = q->size;
	ora_kswr_t r = striped_sw(q, tlen, target, o_del, e_del, o_ins, e_ins, xtra), rr;
	sprof_free(q);
	if ((xtra & ORA_KSW_XSTART) == 0 || ((xtra & ORA_KSW_XSUBO) && r.score < (xtra & 0xffff))) return r;
	rev_bytes(r.qe + 1, query); rev_bytes(r.te + 1, target);       /* ksw.c:356: note tlen (not te+1) is scanned below */
	q = sprof_new(size, r.qe + 1, query, m, mat);
	rr = striped_sw(q, tlen, target, o_del, e_del, o_ins, e_ins, ORA_KSW_XSTOP | r.score);
	rev_bytes(r.qe + 1, query); rev_bytes(r.te + 1, target);
	sprof_free(q);
	if (r.score == rr.score) r.tb = r.te - rr.te, r.qb = r.qe - rr.qe;
	return r;
}
