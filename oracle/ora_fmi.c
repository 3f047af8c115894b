/* ORACLE (test infrastructure only) -- FM-index operations.
 * Restates bwt.c of the reference: Occ (bwt.c:107,169,189), bidirectional
 * extension (bwt.c:262), SMEM search (bwt.c:289), forward-only re-seeding
 * (bwt.c:358) and suffix-array lookup (bwt.c:53,86).
 */
#include <stdlib.h>
#include <string.h>
#include "ora.h"

/* Number of symbols == c among the 16 bases packed in w (2 bits each), for all
 * four c at once, returned as 4 byte-wide counters (c=0 in the low byte).  The
 * reference gets the same numbers from a 256-entry table (bwt.c:42-51,165). */
static inline uint32_t count16(uint32_t w)
{
	uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
	uint32_t n3 = (uint32_t)__builtin_popcount(hi & lo);
	uint32_t n2 = (uint32_t)__builtin_popcount(hi & ~lo);
	uint32_t n1 = (uint32_t)__builtin_popcount(~hi & lo & 0x55555555u);
	uint32_t n0 = 16 - n1 - n2 - n3;
	return n0 | n1 << 8 | n2 << 16 | n3 << 24;
}

/* bwt.c:169 bwt_occ4: counts of A,C,G,T in BWT[0..k] (k inclusive). */
void ora_occ4(const ora_fmi_t *f, uint64_t k, uint64_t cnt[4])
{
	const uint32_t *blk, *w, *stop;
	uint32_t acc = 0, last;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }   /* bwt.c:173 */
	k -= (k >= f->primary);                                                        /* bwt.c:177: '$' is not stored */
	blk = f->bwt + ((k >> 7) << 4);
	memcpy(cnt, blk, 32);
	w = blk + 8;
	stop = w + ((k & 127) >> 4);
	for (; w < stop; ++w) acc += count16(*w);
	last = *w & ~((1u << ((~k & 15) << 1)) - 1);                                   /* bwt.c:183 */
	acc += count16(last) - (uint32_t)(~k & 15);                                    /* phantom A's of the masked tail */
	cnt[0] += acc & 0xff; cnt[1] += acc >> 8 & 0xff; cnt[2] += acc >> 16 & 0xff; cnt[3] += acc >> 24;
}

/* bwt.c:107 bwt_occ.  Equal to occ4(k)[c] for every k in [-1, seq_len]. */
uint64_t ora_occ(const ora_fmi_t *f, uint64_t k, int c)
{
	uint64_t cnt[4];
	if (k == f->seq_len) return f->L2[c + 1] - f->L2[c];
	ora_occ4(f, k, cnt);
	return cnt[c];
}

/* bwt.c:262 bwt_extend. */
void ora_extend(const ora_fmi_t *f, const ora_intv_t *ik, ora_intv_t ok[4], int is_back)
{
	uint64_t tk[4], tl[4];
	int i, a = !is_back, b = is_back;
	ora_occ4(f, ik->x[a] - 1, tk);
	ora_occ4(f, ik->x[a] - 1 + ik->x[2], tl);
	for (i = 0; i < 4; ++i) {
		ok[i].x[a] = f->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[b] = ik->x[b] + (ik->x[a] <= f->primary && ik->x[a] + ik->x[2] - 1 >= f->primary);
	ok[2].x[b] = ok[3].x[b] + ok[3].x[2];
	ok[1].x[b] = ok[2].x[b] + ok[2].x[2];
	ok[0].x[b] = ok[1].x[b] + ok[1].x[2];
}

void ora_set_intv(const ora_fmi_t *f, int c, ora_intv_t *ik)                      /* bwt.h:82 */
{
	ik->x[0] = f->L2[c] + 1;
	ik->x[2] = f->L2[c + 1] - f->L2[c];
	ik->x[1] = f->L2[3 - c] + 1;
	ik->info = 0;
}

static inline void iv_push(ora_intv_v *v, const ora_intv_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 16; v->a = (ora_intv_t*)realloc(v->a, v->m * sizeof(ora_intv_t)); }
	v->a[v->n++] = *p;
}
static void iv_reverse(ora_intv_v *v)
{
	int i;
	for (i = 0; i < v->n >> 1; ++i) { ora_intv_t t = v->a[i]; v->a[i] = v->a[v->n - 1 - i]; v->a[v->n - 1 - i] = t; }
}

/* bwt.c:289 bwt_smem1a.  tmp[0]/tmp[1] are caller-provided scratch vectors. */
int ora_smem1a(const ora_fmi_t *f, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv,
               ora_intv_v *mem, ora_intv_v *tmp[2])
{
	int i, j, c, ret;
	ora_intv_t ik, ok[4];
	ora_intv_v *prev = tmp[0], *curr = tmp[1], *sw;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	ora_set_intv(f, q[x], &ik);
	ik.info = x + 1;
	for (i = x + 1, curr->n = 0; i < len; ++i) {          /* forward extension (bwt.c:304) */
		if (ik.x[2] < max_intv) { iv_push(curr, &ik); break; }
		else if (q[i] < 4) {
			c = 3 - q[i];
			ora_extend(f, &ik, ok, 0);
			if (ok[c].x[2] != ik.x[2]) {
				iv_push(curr, &ik);
				if (ok[c].x[2] < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else { iv_push(curr, &ik); break; }
	}
	if (i == len) iv_push(curr, &ik);
	iv_reverse(curr);
	ret = (int)curr->a[0].info;
	sw = curr; curr = prev; prev = sw;
	for (i = x - 1; i >= -1; --i) {                        /* backward extension (bwt.c:326) */
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, curr->n = 0; j < prev->n; ++j) {
			ora_intv_t *p = &prev->a[j];
			if (c >= 0 && ik.x[2] >= max_intv) ora_extend(f, p, ok, 1);
			if (c < 0 || ik.x[2] < max_intv || ok[c].x[2] < (uint64_t)min_intv) {
				if (curr->n == 0) {
					if (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n - 1].info >> 32) {
						ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
						iv_push(mem, &ik);
					}
				}
			} else if (curr->n == 0 || ok[c].x[2] != curr->a[curr->n - 1].x[2]) {
				ok[c].info = p->info;
				iv_push(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		sw = curr; curr = prev; prev = sw;
	}
	iv_reverse(mem);
	return ret;
}

/* bwt.c:358 bwt_seed_strategy1. */
int ora_seed_strategy1(const ora_fmi_t *f, int len, const uint8_t *q, int x, int min_len, int max_intv, ora_intv_t *mem)
{
	int i, c;
	ora_intv_t ik, ok[4];
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	ora_set_intv(f, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		if (q[i] < 4) {
			c = 3 - q[i];
			ora_extend(f, &ik, ok, 0);
			if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) {
				*mem = ok[c];
				mem->info = (uint64_t)x << 32 | (uint32_t)(i + 1);
				return i + 1;
			}
			ik = ok[c];
		} else return i + 1;
	}
	return len;
}

/* bwt.h:80 bwt_B0: base at row k of the '$'-removed BWT. */
static inline int bwt_base(const ora_fmi_t *f, uint64_t k)
{
	return f->bwt[((k >> 7) << 4) + 8 + ((k & 127) >> 4)] >> ((~k & 15) << 1) & 3;
}

/* bwt.c:53 bwt_invPsi (one LF step). */
static inline uint64_t lf_step(const ora_fmi_t *f, uint64_t k)
{
	uint64_t x = k - (k > f->primary);
	int c = bwt_base(f, x);
	x = f->L2[c] + ora_occ(f, k, c);
	return k == f->primary ? 0 : x;
}

/* bwt.c:86 bwt_sa. */
uint64_t ora_sa(const ora_fmi_t *f, uint64_t k)
{
	uint64_t steps = 0, mask = (uint64_t)f->sa_intv - 1;
	while (k & mask) { ++steps; k = lf_step(f, k); }
	return steps + f->sa[k / f->sa_intv];
}
