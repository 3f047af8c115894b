/* TEST INFRASTRUCTURE ONLY -- never linked into, or called from, the product.
 *
 * Driver around the *reference's own* CPU implementation (the stock bwa 0.7.17
 * path that still lives in /root/reference: bwamem.c, bwt.c, ksw.c, bntseq.c,
 * bwa.c, bwamem_pair.c, bwamem_extra.c ...).  The reference sources are
 * compiled where they lie by oracle/Makefile; only this file is ours.  It is
 * used to (a) pin the C restatement in oracle/ and (b) generate the golden
 * vectors committed under tests/golden/.
 *
 * We #include the reference's bwamem.c into this translation unit (instead of
 * linking bwamem.o) for one reason: mem_collect_intv (bwamem.c:137) is static
 * and the per-stage dump needs it.
 *
 * Sub-commands
 *   index  <in.fa> <prefix>                      bwa_idx_build (bwtindex.c:255)
 *   mem    [-p] [-t N] [-K bases] <prefix> <r1.fq> [r2.fq]   SAM body on stdout
 *   stages <prefix> <reads.fq> <out.bin>         per-read stage dump (SE)
 *   katfm  <prefix> <out.bin> <n> <seed>         known answers: occ4/extend/sa
 *   katksw <out.bin> <n> <seed>                  known answers: ksw_extend2 /
 *                                                ksw_global2 / ksw_align2
 *   katalign <out.bin> <n> <seed>                known answers: ksw_align2 under
 *                                                other matrices / gap costs / xtra
 *
 * Dump format ("i64 records"): a stream of  [tag:i64][n:i64][n x i64].
 */
#include "bwamem.c" /* reference source, found through -I/root/reference */
#include <zlib.h>
#include <sys/time.h>
static double now_s(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + tv.tv_usec * 1e-6; }
#include <unistd.h>
#include "kseq.h"
KSEQ_DECLARE(gzFile)

extern int bwa_idx_build(const char *fa, const char *prefix, int algo_type, int block_size);
#define OPT_T mem_opt_t
#define PES_T mem_pestat_t
#define OPT_FILL_SCMAT(a, b, mat) bwa_fill_scmat((a), (b), (mat))
#define OPT_LOG(x) log(x)
#include "opt_parse.h"

/* ---------------------------------------------------------------- records */
static void rec_write(FILE *fp, int64_t tag, int64_t n, const int64_t *v)
{
	fwrite(&tag, 8, 1, fp); fwrite(&n, 8, 1, fp);
	if (n) fwrite(v, 8, n, fp);
}
typedef struct { int64_t n, m; int64_t *a; } i64v;
static inline void push(i64v *v, int64_t x)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 256; v->a = realloc(v->a, v->m * 8); }
	v->a[v->n++] = x;
}
static inline int64_t f2i(float f) { uint32_t u; memcpy(&u, &f, 4); return (int64_t)u; }

enum { TAG_READ = 100, TAG_INTV = 1, TAG_CHAIN = 2, TAG_CHAIN_FLT = 3, TAG_REGS_PRE = 5, TAG_REGS = 4,
       TAG_OCC4 = 10, TAG_EXTEND = 11, TAG_SA = 12, TAG_KSW_EXT = 20, TAG_KSW_GLB = 21, TAG_KSW_ALN = 22, TAG_KSW_ALN2 = 23 };

static void dump_chains(FILE *fp, int64_t tag, int n, const mem_chain_t *a)
{
	i64v v = {0, 0, 0};
	int i, j;
	push(&v, n);
	for (i = 0; i < n; ++i) {
		const mem_chain_t *c = &a[i];
		push(&v, c->pos); push(&v, c->rid); push(&v, c->is_alt); push(&v, c->w); push(&v, c->kept);
		push(&v, c->first); push(&v, f2i(c->frac_rep)); push(&v, c->n);
		for (j = 0; j < c->n; ++j) {
			push(&v, c->seeds[j].rbeg); push(&v, c->seeds[j].qbeg);
			push(&v, c->seeds[j].len); push(&v, c->seeds[j].score);
		}
	}
	rec_write(fp, tag, v.n, v.a);
	free(v.a);
}

static void dump_regs(FILE *fp, int64_t tag, int n, const mem_alnreg_t *a)
{
	i64v v = {0, 0, 0};
	int i;
	push(&v, n);
	for (i = 0; i < n; ++i) {
		const mem_alnreg_t *p = &a[i];
		push(&v, p->rb); push(&v, p->re); push(&v, p->qb); push(&v, p->qe); push(&v, p->rid);
		push(&v, p->score); push(&v, p->truesc); push(&v, p->sub); push(&v, p->alt_sc); push(&v, p->csub);
		push(&v, p->sub_n); push(&v, p->w); push(&v, p->seedcov); push(&v, p->secondary);
		push(&v, p->secondary_all); push(&v, p->seedlen0); push(&v, p->n_comp); push(&v, p->is_alt);
		push(&v, f2i(p->frac_rep));
	}
	rec_write(fp, tag, v.n, v.a);
	free(v.a);
}

/* ------------------------------------------------------------------- index */
static int main_index(int argc, char **argv)
{
	if (argc < 3) return 1;
	bwa_verbose = 1;
	return bwa_idx_build(argv[1], argv[2], 0, 10000000);
}

/* --------------------------------------------------------------------- mem */
static int main_mem(int argc, char **argv)
{
	mem_opt_t *opt = mem_opt_init();
	int c, n;
	int64_t n_processed = 0;
	double t_align = 0;
	bwaidx_t *idx;
	gzFile f1, f2 = 0;
	kseq_t *ks, *ks2 = 0;
	bseq1_t *seqs;
	optparse_t op;
	bwa_verbose = 1;
	optparse_init(&op, opt);
	while ((c = getopt(argc, argv, OPT_GETOPT_STRING)) >= 0) if (optparse_one(&op, c, optarg)) return 1;
	if (optparse_finish(&op) || optind + 2 > argc) return 1;
	idx = bwa_idx_load(argv[optind], BWA_IDX_ALL);
	if (idx == 0) return 1;
	if (op.ignore_alt) for (c = 0; c < idx->bns->n_seqs; ++c) idx->bns->anns[c].is_alt = 0;
	f1 = gzopen(argv[optind + 1], "r"); ks = kseq_init(f1);
	if (optind + 2 < argc && !op.smart_pe) { f2 = gzopen(argv[optind + 2], "r"); ks2 = kseq_init(f2); opt->flag |= MEM_F_PE; }
	strcpy(bwa_rg_id, op.rg_id);
	{
		int chunk = op.fixed_chunk > 0 ? op.fixed_chunk : opt->chunk_size * opt->n_threads; /* fastmap.c:304 */
		while ((seqs = bseq_read(chunk, &n, ks, ks2)) != 0) {
			int i;
			if (n == 0) { free(seqs); break; }
			if (!op.copy_comment) for (i = 0; i < n; ++i) { free(seqs[i].comment); seqs[i].comment = 0; } /* stock: no -C */
			if (op.align_only) { /* -Z: the first half of mem_process_seqs only (bwamem.c:1225-1232): kt_for(worker1) == mem_align1_core per read */
				extern void kt_for(int n_threads, void (*func)(void*,int,int), void *data, int n);
				worker_t w;
				double t0;
				w.regs = malloc(n * sizeof(mem_alnreg_v));
				w.opt = opt; w.bwt = idx->bwt; w.bns = idx->bns; w.pac = idx->pac; w.seqs = seqs; w.n_processed = n_processed; w.pes = 0;
				w.aux = malloc(opt->n_threads * sizeof(smem_aux_t*));
				for (i = 0; i < opt->n_threads; ++i) w.aux[i] = smem_aux_init();
				t0 = now_s();
				kt_for(opt->n_threads, worker1, &w, (opt->flag & MEM_F_PE) ? n >> 1 : n);
				t_align += now_s() - t0;
				for (i = 0; i < opt->n_threads; ++i) smem_aux_destroy(w.aux[i]);
				free(w.aux);
				for (i = 0; i < n; ++i) { free(w.regs[i].a); seqs[i].sam = 0; }
				free(w.regs);
			} else
			{ double t0 = now_s(); mem_process_seqs(opt, idx->bwt, idx->bns, idx->pac, n_processed, n, seqs, op.has_pes0 ? op.pes : 0); t_align += now_s() - t0; }
			n_processed += n;
			for (i = 0; i < n; ++i) {
				if (seqs[i].sam) fputs(seqs[i].sam, stdout);
				free(seqs[i].name); free(seqs[i].seq); free(seqs[i].qual); free(seqs[i].sam);
			}
			free(seqs);
		}
	}
	fprintf(stderr, "[bwaref] aligned %lld reads in %.3f s with %d threads (mem_process_seqs only)\n", (long long)n_processed, t_align, opt->n_threads);
	kseq_destroy(ks); gzclose(f1);
	if (ks2) { kseq_destroy(ks2); gzclose(f2); }
	bwa_idx_destroy(idx);
	free(opt);
	return 0;
}

/* ------------------------------------------------------------------ stages */
static int main_stages(int argc, char **argv)
{
	mem_opt_t *opt = mem_opt_init();
	bwaidx_t *idx;
	gzFile f1;
	kseq_t *ks;
	FILE *out;
	int64_t id = 0;
	optparse_t op;
	int c;
	bwa_verbose = 1;
	optparse_init(&op, opt);
	while ((c = getopt(argc, argv, OPT_GETOPT_STRING)) >= 0) if (optparse_one(&op, c, optarg)) return 1;
	if (optparse_finish(&op) || optind + 3 > argc) return 1;
	argv += optind - 1;
	idx = bwa_idx_load(argv[1], BWA_IDX_ALL);
	if (idx == 0) return 1;
	if (op.ignore_alt) for (c = 0; c < idx->bns->n_seqs; ++c) idx->bns->anns[c].is_alt = 0;
	f1 = gzopen(argv[2], "r"); ks = kseq_init(f1);
	out = fopen(argv[3], "wb");
	while (kseq_read(ks) >= 0) {
		int i, l_seq = ks->seq.l;
		char *seq = malloc(l_seq + 1);
		smem_aux_t *aux = smem_aux_init();
		mem_chain_v chn;
		mem_alnreg_v regs;
		int64_t hdr[2];
		i64v v = {0, 0, 0};
		memcpy(seq, ks->seq.s, l_seq + 1);
		for (i = 0; i < l_seq; ++i) seq[i] = seq[i] < 4 ? seq[i] : nst_nt4_table[(int)seq[i]]; /* bwamem.c:1067 */
		hdr[0] = id++; hdr[1] = l_seq;
		rec_write(out, TAG_READ, 2, hdr);
		/* stage 1: intervals (bwamem.c:137) */
		if (l_seq >= opt->min_seed_len) { /* mem_chain's guard, bwamem.c:267 */
			mem_collect_intv(opt, idx->bwt, l_seq, (uint8_t*)seq, aux);
			for (i = 0; i < aux->mem.n; ++i) {
				push(&v, aux->mem.a[i].x[0]); push(&v, aux->mem.a[i].x[1]);
				push(&v, aux->mem.a[i].x[2]); push(&v, aux->mem.a[i].info);
			}
		}
		rec_write(out, TAG_INTV, v.n, v.a);
		free(v.a);
		/* stage 2: chains (bwamem.c:258) */
		chn = mem_chain(opt, idx->bwt, idx->bns, l_seq, (uint8_t*)seq, aux);
		dump_chains(out, TAG_CHAIN, chn.n, chn.a);
		/* stage 3: filtered chains (bwamem.c:334, 605) */
		chn.n = mem_chain_flt(opt, chn.n, chn.a);
		mem_flt_chained_seeds(opt, idx->bns, idx->pac, l_seq, (uint8_t*)seq, chn.n, chn.a);
		dump_chains(out, TAG_CHAIN_FLT, chn.n, chn.a);
		/* stage 4: extension (bwamem.c:639) */
		kv_init(regs);
		for (i = 0; i < chn.n; ++i) {
			mem_chain2aln(opt, idx->bns, idx->pac, l_seq, (uint8_t*)seq, &chn.a[i], &regs);
			free(chn.a[i].seeds);
		}
		free(chn.a);
		dump_regs(out, TAG_REGS_PRE, regs.n, regs.a);
		/* stage 5: dedup/patch (bwamem.c:444) + is_alt (bwamem.c:1091) */
		regs.n = mem_sort_dedup_patch(opt, idx->bns, idx->pac, (uint8_t*)seq, regs.n, regs.a);
		for (i = 0; i < regs.n; ++i)
			if (regs.a[i].rid >= 0 && idx->bns->anns[regs.a[i].rid].is_alt) regs.a[i].is_alt = 1;
		dump_regs(out, TAG_REGS, regs.n, regs.a);
		free(regs.a);
		smem_aux_destroy(aux);
		free(seq);
	}
	fclose(out);
	kseq_destroy(ks); gzclose(f1);
	bwa_idx_destroy(idx);
	free(opt);
	return 0;
}

/* ------------------------------------------------------- known answers: FM */
static uint64_t rng_state;
static inline uint64_t rng(void)
{ /* splitmix64 */
	uint64_t z = (rng_state += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}

static int main_katfm(int argc, char **argv)
{
	bwaidx_t *idx;
	FILE *out;
	int i, n;
	if (argc < 5) return 1;
	bwa_verbose = 1;
	idx = bwa_idx_load(argv[1], BWA_IDX_BWT);
	out = fopen(argv[2], "wb");
	n = atoi(argv[3]); rng_state = strtoull(argv[4], 0, 10);
	for (i = 0; i < n; ++i) {
		const bwt_t *bwt = idx->bwt;
		bwtint_t k = i == 0 ? (bwtint_t)-1 : i == 1 ? bwt->seq_len : i == 2 ? bwt->primary : i == 3 ? bwt->primary - 1 :
			i == 4 ? 0 : rng() % (bwt->seq_len + 1);
		bwtint_t cnt[4];
		int64_t r[8];
		bwt_occ4(bwt, k, cnt);
		r[0] = k; r[1] = cnt[0]; r[2] = cnt[1]; r[3] = cnt[2]; r[4] = cnt[3];
		rec_write(out, TAG_OCC4, 5, r);
		if (k != (bwtint_t)-1) {
			r[1] = bwt_sa(bwt, k);
			rec_write(out, TAG_SA, 2, r);
		}
	}
	for (i = 0; i < n; ++i) { /* random walks of bwt_extend in both directions */
		const bwt_t *bwt = idx->bwt;
		bwtintv_t ik, ok[4];
		int step, c0 = rng() & 3;
		bwt_set_intv(bwt, c0, ik);
		for (step = 0; step < 40 && ik.x[2] > 0; ++step) {
			int is_back = rng() & 1, c = rng() & 3, j;
			int64_t r[3 + 1 + 12];
			bwt_extend(bwt, &ik, ok, is_back);
			r[0] = ik.x[0]; r[1] = ik.x[1]; r[2] = ik.x[2]; r[3] = is_back;
			for (j = 0; j < 4; ++j) { r[4 + j*3] = ok[j].x[0]; r[5 + j*3] = ok[j].x[1]; r[6 + j*3] = ok[j].x[2]; }
			rec_write(out, TAG_EXTEND, 16, r);
			ik = ok[c];
		}
	}
	fclose(out);
	bwa_idx_destroy(idx);
	return 0;
}

/* ------------------------------------------------------ known answers: ksw */
static void rand_pair(int qlen, int tlen, int err_pct, uint8_t *q, uint8_t *t)
{ /* target = mutated copy of query, so alignments are non-trivial */
	int i, j = 0;
	for (i = 0; i < qlen; ++i) q[i] = rng() & 3;
	for (i = 0; i < tlen; ++i) {
		int r = rng() % 100;
		if (j >= qlen) { t[i] = rng() & 3; continue; }
		if (r < err_pct) { /* substitution / insertion / deletion, 1/3 each */
			int kind = rng() % 3;
			if (kind == 0) t[i] = (q[j] + 1 + rng() % 3) & 3, ++j;
			else if (kind == 1) t[i] = rng() & 3;
			else { j += 1 + rng() % 3; t[i] = j < qlen ? q[j] : (rng() & 3); ++j; }
		} else t[i] = q[j++];
		if ((rng() & 255) == 0) t[i] = 4; /* ambiguous base */
	}
}

static int main_katksw(int argc, char **argv)
{
	FILE *out;
	int i, n;
	int8_t mat[25];
	if (argc < 4) return 1;
	out = fopen(argv[1], "wb");
	n = atoi(argv[2]); rng_state = strtoull(argv[3], 0, 10);
	bwa_fill_scmat(1, 4, mat);
	for (i = 0; i < n; ++i) {
		int qlen = 1 + rng() % 250, tlen = qlen + (int)(rng() % 60) - 20, err = rng() % 12;
		int w = (rng() & 3) == 0 ? 5 + rng() % 20 : 100 << (rng() & 1);
		int h0 = 19 + rng() % 100, zdrop = (rng() & 7) == 0 ? 0 : 100, end_bonus = (rng() & 1) ? 5 : 0;
		int o_del = 6, e_del = 1, o_ins = 6, e_ins = 1;
		uint8_t *q, *t;
		i64v v = {0, 0, 0};
		int j, qle, tle, gtle, gscore, max_off, sc;
		if (tlen < 1) tlen = 1;
		if ((rng() & 15) == 0) { o_del = 4; e_del = 2; o_ins = 7; e_ins = 1; }
		q = malloc(qlen); t = malloc(tlen);
		rand_pair(qlen, tlen, err, q, t);
		/* inputs */
		push(&v, qlen); push(&v, tlen); push(&v, w); push(&v, h0); push(&v, zdrop); push(&v, end_bonus);
		push(&v, o_del); push(&v, e_del); push(&v, o_ins); push(&v, e_ins);
		for (j = 0; j < qlen; ++j) push(&v, q[j]);
		for (j = 0; j < tlen; ++j) push(&v, t[j]);
		/* ksw_extend2 (ksw.c:380) */
		sc = ksw_extend2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0, &qle, &tle, &gtle, &gscore, &max_off);
		push(&v, sc); push(&v, qle); push(&v, tle); push(&v, gtle); push(&v, gscore); push(&v, max_off);
		rec_write(out, TAG_KSW_EXT, v.n, v.a);
		/* ksw_global2 (ksw.c:504) with CIGAR; band must admit the length difference */
		{
			int n_cigar = 0, wg = w > abs(tlen - qlen) + 3 ? w : abs(tlen - qlen) + 3;
			uint32_t *cigar = 0;
			v.n = 10 + qlen + tlen; v.a[2] = wg;
			sc = ksw_global2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, wg, &n_cigar, &cigar);
			push(&v, sc); push(&v, n_cigar);
			for (j = 0; j < n_cigar; ++j) push(&v, cigar[j]);
			rec_write(out, TAG_KSW_GLB, v.n, v.a);
			free(cigar);
		}
		/* ksw_align2 (ksw.c:343) as mem_matesw calls it (bwamem_pair.c:166) */
		{
			kswr_t r;
			int xtra = KSW_XSUBO | KSW_XSTART | ((qlen * 1 < 250) ? KSW_XBYTE : 0) | (19 * 1);
			v.n = 10 + qlen + tlen;
			r = ksw_align2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, xtra, 0);
			push(&v, r.score); push(&v, r.te); push(&v, r.qe); push(&v, r.score2); push(&v, r.te2);
			push(&v, r.tb); push(&v, r.qb);
			rec_write(out, TAG_KSW_ALN, v.n, v.a);
		}
		free(v.a); free(q); free(t);
	}
	fclose(out);
	return 0;
}

/* ksw_align2 (ksw.c:343) beyond the one call shape of katksw: word kernel as mem_seed_sw calls it (KSW_XSTART only,
 * bwamem.c:601), byte / word kernels with a sub-optimal threshold as mem_matesw does (bwamem_pair.c:165), under several
 * scoring matrices and gap costs.  Record: qlen tlen xtra o_del e_del o_ins e_ins mat[25] q[qlen] t[tlen] | 7 results. */
static int main_katalign(int argc, char **argv)
{
	static const int ab[4][2] = { {1, 4}, {2, 3}, {1, 1}, {3, 9} };
	static const int gaps[4][4] = { {6, 1, 6, 1}, {4, 2, 7, 1}, {1, 1, 1, 1}, {16, 1, 16, 1} };
	FILE *out;
	int i, n;
	if (argc < 4) return 1;
	out = fopen(argv[1], "wb");
	n = atoi(argv[2]); rng_state = strtoull(argv[3], 0, 10);
	for (i = 0; i < n; ++i) {
		const int *sc = ab[rng() % 4], *g = gaps[rng() % 4];
		int kind = rng() % 3, a = sc[0];
		int qlen = kind == 0 ? 1 + rng() % 199 : 1 + rng() % 400, tlen, err = rng() % 15, xtra, j;
		int8_t mat[25];
		uint8_t *q, *t;
		i64v v = {0, 0, 0};
		kswr_t r;
		if (kind == 0) { tlen = 1 + rng() % 199; xtra = KSW_XSTART; }                       /* mem_seed_sw */
		else {
			tlen = qlen + (int)(rng() % 500);
			xtra = KSW_XSUBO | KSW_XSTART | ((qlen * a < 250) ? KSW_XBYTE : 0) | (19 * a); /* mem_matesw */
			if (kind == 2 && (rng() & 1)) xtra &= ~KSW_XBYTE;                             /* word kernel on short queries too */
		}
		bwa_fill_scmat(a, sc[1], mat);
		q = malloc(qlen); t = malloc(tlen);
		if (kind == 0) rand_pair(qlen, tlen, err, q, t);
		else { /* the query sits somewhere inside a longer random target, mutated */
			int off = rng() % (tlen - qlen + 1);
			for (j = 0; j < tlen; ++j) t[j] = rng() & 3;
			rand_pair(qlen, qlen, err, q, t + off);
			if ((rng() & 3) == 0) for (j = 0; j < qlen && off + qlen + 40 + j < tlen; ++j) t[off + qlen + 40 + j] = q[j]; /* a second copy */
		}
		push(&v, qlen); push(&v, tlen); push(&v, xtra); push(&v, g[0]); push(&v, g[1]); push(&v, g[2]); push(&v, g[3]);
		for (j = 0; j < 25; ++j) push(&v, mat[j]);
		for (j = 0; j < qlen; ++j) push(&v, q[j]);
		for (j = 0; j < tlen; ++j) push(&v, t[j]);
		r = ksw_align2(qlen, q, tlen, t, 5, mat, g[0], g[1], g[2], g[3], xtra, 0);
		push(&v, r.score); push(&v, r.te); push(&v, r.qe); push(&v, r.score2); push(&v, r.te2); push(&v, r.tb); push(&v, r.qb);
		rec_write(out, TAG_KSW_ALN2, v.n, v.a);
		free(v.a); free(q); free(t);
	}
	fclose(out);
	return 0;
}

/* readfq <chunk_bases> <in1> [in2]: the batches the reference's bseq_read (bwa.c:191) cuts, one record per line as
 * name TAB comment-or-* TAB seq TAB qual-or-*, a line "#batch <n>" in front of every batch -- golden vectors for the product's
 * FASTA/FASTQ reader (csrc/fastq_reader.cpp) */
static int main_readfq(int argc, char **argv)
{
	gzFile f1, f2 = 0;
	kseq_t *ks, *ks2 = 0;
	bseq1_t *seqs;
	int n, i, chunk;
	if (argc < 3) return 1;
	chunk = atoi(argv[1]);
	f1 = gzopen(argv[2], "r"); ks = kseq_init(f1);
	if (argc > 3) { f2 = gzopen(argv[3], "r"); ks2 = kseq_init(f2); }
	while ((seqs = bseq_read(chunk, &n, ks, ks2)) != 0) {
		printf("#batch %d\n", n);
		for (i = 0; i < n; ++i) {
			printf("%s\t%s\t%s\t%s\n", seqs[i].name, seqs[i].comment ? seqs[i].comment : "*", seqs[i].seq, seqs[i].qual ? seqs[i].qual : "*");
			free(seqs[i].name); free(seqs[i].comment); free(seqs[i].seq); free(seqs[i].qual);
		}
		free(seqs);
		if (n == 0) break;
	}
	kseq_destroy(ks); gzclose(f1);
	if (ks2) { kseq_destroy(ks2); gzclose(f2); }
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: bwaref <index|mem|stages|katfm|katksw> ...\n");
		return 1;
	}
	if (strcmp(argv[1], "index") == 0) return main_index(argc - 1, argv + 1);
	if (strcmp(argv[1], "mem") == 0) return main_mem(argc - 1, argv + 1);
	if (strcmp(argv[1], "stages") == 0) return main_stages(argc - 1, argv + 1);
	if (strcmp(argv[1], "katfm") == 0) return main_katfm(argc - 1, argv + 1);
	if (strcmp(argv[1], "katksw") == 0) return main_katksw(argc - 1, argv + 1);
	if (strcmp(argv[1], "katalign") == 0) return main_katalign(argc - 1, argv + 1);
	if (strcmp(argv[1], "readfq") == 0) return main_readfq(argc - 1, argv + 1);
	fprintf(stderr, "unknown sub-command '%s'\n", argv[1]);
	return 1;
}
