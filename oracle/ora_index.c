/* ORACLE (test infrastructure only) -- index files and reference-sequence access.
 * Formats follow what the reference writes: .bwt (bwt.c:385-393), .sa
 * (bwt.c:396-407), .pac (bntseq.c:314-326), .ann/.amb (bntseq.c:65-94),
 * optional .alt name list (bntseq.c:178-209).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "ora.h"

const uint8_t ora_nt4_table[256] = {   /* bntseq.c:46: ACGT (either case) -> 0..3, '-' -> 5, everything else 4 */
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,5,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4
};

static FILE *must_open(const char *prefix, const char *ext, const char *mode, int optional)
{
	char fn[4096];
	FILE *fp;
	snprintf(fn, sizeof fn, "%s%s", prefix, ext);
	fp = fopen(fn, mode);
	if (!fp && !optional) { fprintf(stderr, "[ora] cannot open %s\n", fn); exit(1); }
	return fp;
}
static void must_read(void *p, size_t sz, size_t n, FILE *fp)
{
	if (fread(p, sz, n, fp) != n) { fprintf(stderr, "[ora] short read\n"); exit(1); }
}

static ora_fmi_t *load_fmi(const char *prefix)
{
	ora_fmi_t *f = (ora_fmi_t*)calloc(1, sizeof(*f));
	FILE *fp = must_open(prefix, ".bwt", "rb", 0);
	long sz;
	uint64_t hdr[5], v;
	fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, 0, SEEK_SET);
	f->n_words = (sz - 40) >> 2;
	must_read(&f->primary, 8, 1, fp);
	must_read(f->L2 + 1, 8, 4, fp);
	f->bwt = (uint32_t*)malloc(f->n_words * 4 + 64);
	must_read(f->bwt, 4, f->n_words, fp);
	f->seq_len = f->L2[4];
	fclose(fp);
	fp = must_open(prefix, ".sa", "rb", 0);
	must_read(hdr, 8, 5, fp);                     /* primary + 4 skipped words (bwt.c:428-430) */
	assert(hdr[0] == f->primary);
	must_read(&v, 8, 1, fp); f->sa_intv = (int)v;
	must_read(&v, 8, 1, fp); assert(v == f->seq_len);
	f->n_sa = (f->seq_len + f->sa_intv) / f->sa_intv;
	f->sa = (uint64_t*)malloc(f->n_sa * 8);
	f->sa[0] = (uint64_t)-1;
	must_read(f->sa + 1, 8, f->n_sa - 1, fp);
	fclose(fp);
	return f;
}

static ora_ref_t *load_ref(const char *prefix)
{
	ora_ref_t *r = (ora_ref_t*)calloc(1, sizeof(*r));
	FILE *fp = must_open(prefix, ".ann", "r", 0);
	char buf[8192];
	long long xx;
	int i, n_seqs;
	if (fscanf(fp, "%lld%d%u", &xx, &r->n_seqs, &r->seed) != 3) goto bad;
	r->l_pac = xx;
	r->anns = (ora_ann_t*)calloc(r->n_seqs, sizeof(ora_ann_t));
	for (i = 0; i < r->n_seqs; ++i) {
		ora_ann_t *p = &r->anns[i];
		char *q = buf;
		int c;
		if (fscanf(fp, "%u%8191s", &p->gi, buf) != 2) goto bad;
		p->name = strdup(buf);
		while (q - buf < (long)sizeof(buf) - 1 && (c = fgetc(fp)) != '\n' && c != EOF) *q++ = c;
		while (c != '\n' && c != EOF) c = fgetc(fp);
		*q = 0;
		p->anno = (q - buf > 1 && strcmp(buf, " (null)") != 0) ? strdup(buf + 1) : strdup("");
		if (fscanf(fp, "%lld%d%d", &xx, &p->len, &p->n_ambs) != 3) goto bad;
		p->offset = xx;
	}
	fclose(fp);
	fp = must_open(prefix, ".amb", "r", 0);
	if (fscanf(fp, "%lld%d%d", &xx, &n_seqs, &r->n_holes) != 3) goto bad;
	r->ambs = r->n_holes ? (ora_amb_t*)calloc(r->n_holes, sizeof(ora_amb_t)) : 0;
	for (i = 0; i < r->n_holes; ++i) {
		if (fscanf(fp, "%lld%d%8191s", &xx, &r->ambs[i].len, buf) != 3) goto bad;
		r->ambs[i].offset = xx; r->ambs[i].amb = buf[0];
	}
	fclose(fp);
	fp = must_open(prefix, ".pac", "rb", 0);
	r->pac = (uint8_t*)calloc(r->l_pac / 4 + 1, 1);
	must_read(r->pac, 1, r->l_pac / 4 + 1, fp);  /* bwa.c:421-422 */
	fclose(fp);
	if ((fp = must_open(prefix, ".alt", "r", 1)) != 0) {   /* bntseq.c:178-209: first column = contig name */
		while (fgets(buf, sizeof buf, fp)) {
			char *e = buf;
			if (buf[0] == '@') continue;
			while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			for (i = 0; i < r->n_seqs; ++i)
				if (strcmp(r->anns[i].name, buf) == 0) { r->anns[i].is_alt = 1; break; }
		}
		fclose(fp);
	}
	return r;
bad:
	fprintf(stderr, "[ora] parse error in %s.ann/.amb\n", prefix);
	exit(1);
}

ora_index_t *ora_index_load(const char *prefix)
{
	ora_index_t *idx = (ora_index_t*)calloc(1, sizeof(*idx));
	idx->fmi = load_fmi(prefix);
	idx->ref = load_ref(prefix);
	return idx;
}

void ora_index_destroy(ora_index_t *idx)
{
	int i;
	if (!idx) return;
	free(idx->fmi->bwt); free(idx->fmi->sa); free(idx->fmi);
	for (i = 0; i < idx->ref->n_seqs; ++i) { free(idx->ref->anns[i].name); free(idx->ref->anns[i].anno); }
	free(idx->ref->anns); free(idx->ref->ambs); free(idx->ref->pac); free(idx->ref);
	free(idx);
}

int ora_pos2rid(const ora_ref_t *r, int64_t pos_f)     /* bntseq.c:354 */
{
	int left = 0, mid = 0, right = r->n_seqs;
	if (pos_f >= r->l_pac) return -1;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= r->anns[mid].offset) {
			if (mid == r->n_seqs - 1) break;
			if (pos_f < r->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

int ora_intv2rid(const ora_ref_t *r, int64_t rb, int64_t re)   /* bntseq.c:370 */
{
	int is_rev, rid_b, rid_e;
	if (rb < r->l_pac && re > r->l_pac) return -2;
	assert(rb <= re);
	rid_b = ora_pos2rid(r, ora_depos(r, rb, &is_rev));
	rid_e = rb < re ? ora_pos2rid(r, ora_depos(r, re - 1, &is_rev)) : rid_b;
	return rid_b == rid_e ? rid_b : -1;
}

#define PAC_AT(pac, l) ((pac)[(l) >> 2] >> ((~(l) & 3) << 1) & 3)

uint8_t *ora_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len)   /* bntseq.c:403 */
{
	uint8_t *seq = 0;
	if (end < beg) { int64_t t = beg; beg = end; end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	if (beg >= l_pac || end <= l_pac) {
		int64_t k, l = 0;
		*len = end - beg;
		seq = (uint8_t*)malloc(end - beg > 0 ? end - beg : 1);
		if (beg >= l_pac) {
			int64_t beg_f = (l_pac << 1) - 1 - end, end_f = (l_pac << 1) - 1 - beg;
			for (k = end_f; k > beg_f; --k) seq[l++] = 3 - PAC_AT(pac, k);
		} else for (k = beg; k < end; ++k) seq[l++] = PAC_AT(pac, k);
	} else *len = 0;
	return seq;
}

uint8_t *ora_fetch_seq(const ora_ref_t *r, int64_t *beg, int64_t mid, int64_t *end, int *rid)   /* bntseq.c:426 */
{
	int64_t far_beg, far_end, len;
	int is_rev;
	uint8_t *seq;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	assert(*beg <= mid && mid < *end);
	*rid = ora_pos2rid(r, ora_depos(r, mid, &is_rev));
	far_beg = r->anns[*rid].offset;
	far_end = far_beg + r->anns[*rid].len;
	if (is_rev) {
		int64_t t = far_beg;
		far_beg = (r->l_pac << 1) - far_end;
		far_end = (r->l_pac << 1) - t;
	}
	*beg = *beg > far_beg ? *beg : far_beg;
	*end = *end < far_end ? *end : far_end;
	seq = ora_get_seq(r->l_pac, r->pac, *beg, *end, &len);
	assert(seq && *end - *beg == len);
	return seq;
}
