/* ORACLE (test infrastructure only) -- command-line front end of the C
 * restatement.  Mirrors oracle/ref_driver.c so tests can diff the two:
 *   bwa_oracle mem    [-p] [-t N] [-K bases] [-a] <prefix> <r1.fq> [r2.fq]   SAM body on stdout
 *   bwa_oracle stages <prefix> <reads.fq> <out.bin>                          per-read stage dump
 * Batching follows bseq_read (bwa.c:191): reads are taken until the batch holds
 * >= chunk bases and an even number of reads.
 */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <ctype.h>
#include <zlib.h>
#include <sys/time.h>
static double now_s(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + tv.tv_usec * 1e-6; }
#include <math.h>
#include "ora.h"
#define OPT_T ora_opt_t
#define PES_T ora_pestat_t
#define OPT_FILL_SCMAT(a, b, mat) ora_fill_scmat((a), (b), (mat))
#define OPT_LOG(x) log(x)
#include "opt_parse.h"

/* ---- minimal FASTA/FASTQ reader with kseq.h's field semantics ---- */
typedef struct { gzFile fp; char *line; size_t m; int peeked; } fq_t;
static int fq_getline(fq_t *f)
{
	size_t l = 0;
	if (f->peeked) { f->peeked = 0; return 1; }
	for (;;) {
		if (l + 4096 > f->m) { f->m = f->m ? f->m << 1 : 8192; f->line = (char*)realloc(f->line, f->m); }
		if (!gzgets(f->fp, f->line + l, (int)(f->m - l))) { if (l == 0) return 0; break; }
		l += strlen(f->line + l);
		if (l && f->line[l-1] == '\n') break;
	}
	while (l && (f->line[l-1] == '\n' || f->line[l-1] == '\r')) f->line[--l] = 0;
	return 1;
}
static int fq_read(fq_t *f, ora_read_t *r)
{
	char *p, *q;
	size_t l_seq = 0, m_seq = 0;
	memset(r, 0, sizeof *r);
	do { if (!fq_getline(f)) return 0; } while (f->line[0] != '>' && f->line[0] != '@');
	p = f->line + 1;
	for (q = p; *q && !isspace((unsigned char)*q); ++q);
	r->name = strndup(p, q - p);
	while (*q && isspace((unsigned char)*q)) ++q;
	if (*q) r->comment = strdup(q);
	while (fq_getline(f)) {
		size_t l;
		if (f->line[0] == '+' || f->line[0] == '>' || f->line[0] == '@') { if (f->line[0] != '+') f->peeked = 1; break; }
		l = strlen(f->line);
		if (l_seq + l + 1 > m_seq) { m_seq = (l_seq + l + 1) * 2; r->seq = (char*)realloc(r->seq, m_seq); }
		memcpy(r->seq + l_seq, f->line, l); l_seq += l;
	}
	if (!r->seq) r->seq = (char*)calloc(1, 1);
	r->seq[l_seq] = 0; r->l_seq = (int)l_seq;
	if (!f->peeked && f->line && f->line[0] == '+') {       /* quality: as many characters as bases */
		size_t l_q = 0;
		r->qual = (char*)malloc(l_seq + 1);
		while (l_q < l_seq && fq_getline(f)) {
			size_t l = strlen(f->line);
			if (l_q + l > l_seq) l = l_seq - l_q;
			memcpy(r->qual + l_q, f->line, l); l_q += l;
		}
		r->qual[l_q] = 0;
	}
	{ size_t l = strlen(r->name);                           /* bwa.c:73 trim_readno */
	  if (l > 2 && r->name[l-2] == '/' && isdigit((unsigned char)r->name[l-1])) r->name[l-2] = 0; }
	return 1;
}

static ora_read_t *read_batch(int chunk, int *n_, fq_t *f1, fq_t *f2)   /* bwa.c:191 bseq_read */
{
	int size = 0, m = 0, n = 0;
	ora_read_t *seqs = 0, r;
	while (fq_read(f1, &r)) {
		ora_read_t r2;
		if (f2 && !fq_read(f2, &r2)) break;
		if (n + 2 > m) { m = m ? m << 1 : 256; seqs = (ora_read_t*)realloc(seqs, m * sizeof(ora_read_t)); }
		r.id = n; seqs[n++] = r; size += r.l_seq;
		if (f2) { r2.id = n; seqs[n++] = r2; size += r2.l_seq; }
		if (size >= chunk && (n & 1) == 0) break;
	}
	*n_ = n;
	return seqs;
}

/* ---- record dump, same layout as ref_driver.c ---- */
static void rec_write(FILE *fp, int64_t tag, int64_t n, const int64_t *v) { fwrite(&tag, 8, 1, fp); fwrite(&n, 8, 1, fp); if (n) fwrite(v, 8, n, fp); }
typedef struct { int64_t n, m; int64_t *a; } i64v;
static inline void push(i64v *v, int64_t x) { if (v->n == v->m) { v->m = v->m ? v->m << 1 : 256; v->a = (int64_t*)realloc(v->a, v->m * 8); } v->a[v->n++] = x; }
static inline int64_t f2i(float f) { uint32_t u; memcpy(&u, &f, 4); return (int64_t)u; }
enum { TAG_READ = 100, TAG_INTV = 1, TAG_CHAIN = 2, TAG_CHAIN_FLT = 3, TAG_REGS_PRE = 5, TAG_REGS = 4 };

static void dump_chains(FILE *fp, int64_t tag, int n, const ora_chain_t *a)
{
	i64v v = { 0, 0, 0 };
	int i, j;
	push(&v, n);
	for (i = 0; i < n; ++i) {
		const ora_chain_t *c = &a[i];
		push(&v, c->pos); push(&v, c->rid); push(&v, c->is_alt); push(&v, c->w); push(&v, c->kept);
		push(&v, c->first); push(&v, f2i(c->frac_rep)); push(&v, c->n);
		for (j = 0; j < c->n; ++j) { push(&v, c->seeds[j].rbeg); push(&v, c->seeds[j].qbeg); push(&v, c->seeds[j].len); push(&v, c->seeds[j].score); }
	}
	rec_write(fp, tag, v.n, v.a); free(v.a);
}
static void dump_regs(FILE *fp, int64_t tag, int n, const ora_reg_t *a)
{
	i64v v = { 0, 0, 0 };
	int i;
	push(&v, n);
	for (i = 0; i < n; ++i) {
		const ora_reg_t *p = &a[i];
		push(&v, p->rb); push(&v, p->re); push(&v, p->qb); push(&v, p->qe); push(&v, p->rid);
		push(&v, p->score); push(&v, p->truesc); push(&v, p->sub); push(&v, p->alt_sc); push(&v, p->csub);
		push(&v, p->sub_n); push(&v, p->w); push(&v, p->seedcov); push(&v, p->secondary);
		push(&v, p->secondary_all); push(&v, p->seedlen0); push(&v, p->n_comp); push(&v, p->is_alt);
		push(&v, f2i(p->frac_rep));
	}
	rec_write(fp, tag, v.n, v.a); free(v.a);
}

static int main_stages(int argc, char **argv)
{
	ora_opt_t opt;
	ora_index_t *idx;
	fq_t f = { 0, 0, 0, 0 };
	FILE *out;
	ora_read_t r;
	int64_t id = 0;
	optparse_t op;
	int c;
	ora_opt_init(&opt);
	optparse_init(&op, &opt);
	while ((c = getopt(argc, argv, OPT_GETOPT_STRING)) >= 0) if (optparse_one(&op, c, optarg)) return 1;
	if (optparse_finish(&op) || optind + 3 > argc) return 1;
	argv += optind - 1;
	idx = ora_index_load(argv[1]);
	if (op.ignore_alt) for (c = 0; c < idx->ref->n_seqs; ++c) idx->ref->anns[c].is_alt = 0;
	f.fp = gzopen(argv[2], "r");
	out = fopen(argv[3], "wb");
	while (fq_read(&f, &r)) {
		int i, l_seq = r.l_seq;
		char *seq = r.seq;
		ora_aux_t *aux = ora_aux_new();
		ora_chain_v chn;
		ora_reg_v regs = { 0, 0, 0 };
		int64_t hdr[2];
		i64v v = { 0, 0, 0 };
		for (i = 0; i < l_seq; ++i) seq[i] = seq[i] < 4 ? seq[i] : ora_nt4_table[(uint8_t)seq[i]];
		hdr[0] = id++; hdr[1] = l_seq;
		rec_write(out, TAG_READ, 2, hdr);
		if (l_seq >= opt.min_seed_len) {
			ora_collect_intv(&opt, idx->fmi, l_seq, (uint8_t*)seq, aux);
			for (i = 0; i < aux->mem.n; ++i) { push(&v, aux->mem.a[i].x[0]); push(&v, aux->mem.a[i].x[1]); push(&v, aux->mem.a[i].x[2]); push(&v, aux->mem.a[i].info); }
		}
		rec_write(out, TAG_INTV, v.n, v.a); free(v.a);
		chn = ora_chain(&opt, idx, l_seq, (uint8_t*)seq, aux);
		dump_chains(out, TAG_CHAIN, chn.n, chn.a);
		chn.n = ora_chain_flt(&opt, chn.n, chn.a);
		ora_flt_chained_seeds(&opt, idx->ref, l_seq, (uint8_t*)seq, chn.n, chn.a);
		dump_chains(out, TAG_CHAIN_FLT, chn.n, chn.a);
		for (i = 0; i < chn.n; ++i) { ora_chain2aln(&opt, idx->ref, l_seq, (uint8_t*)seq, &chn.a[i], &regs); free(chn.a[i].seeds); }
		free(chn.a);
		dump_regs(out, TAG_REGS_PRE, regs.n, regs.a);
		regs.n = ora_sort_dedup_patch(&opt, idx->ref, (uint8_t*)seq, regs.n, regs.a);
		for (i = 0; i < regs.n; ++i)
			if (regs.a[i].rid >= 0 && idx->ref->anns[regs.a[i].rid].is_alt) regs.a[i].is_alt = 1;
		dump_regs(out, TAG_REGS, regs.n, regs.a);
		free(regs.a); ora_aux_free(aux);
		free(r.name); free(r.comment); free(r.seq); free(r.qual);
	}
	fclose(out); gzclose(f.fp);
	ora_index_destroy(idx);
	return 0;
}

static int main_mem(int argc, char **argv)
{
	ora_opt_t opt;
	int c, n;
	int64_t n_processed = 0;
	double t_align = 0;
	ora_index_t *idx;
	fq_t f1 = { 0, 0, 0, 0 }, f2 = { 0, 0, 0, 0 };
	ora_read_t *seqs;
	optparse_t op;
	ora_opt_init(&opt);
	optparse_init(&op, &opt);
	while ((c = getopt(argc, argv, OPT_GETOPT_STRING)) >= 0) if (optparse_one(&op, c, optarg)) return 1;
	if (optparse_finish(&op) || optind + 2 > argc) return 1;
	idx = ora_index_load(argv[optind]);
	if (op.ignore_alt) for (c = 0; c < idx->ref->n_seqs; ++c) idx->ref->anns[c].is_alt = 0;
	f1.fp = gzopen(argv[optind + 1], "r");
	if (optind + 2 < argc && !op.smart_pe) { f2.fp = gzopen(argv[optind + 2], "r"); opt.flag |= ORA_F_PE; }
	strcpy(ora_rg_id, op.rg_id);
	{
		int chunk = op.fixed_chunk > 0 ? op.fixed_chunk : opt.chunk_size * opt.n_threads;   /* fastmap.c:304 */
		while ((seqs = read_batch(chunk, &n, &f1, f2.fp ? &f2 : 0)) != 0) {
			int i;
			if (n == 0) { free(seqs); break; }
			if (!op.copy_comment) for (i = 0; i < n; ++i) { free(seqs[i].comment); seqs[i].comment = 0; }     /* stock behaviour without -C */
			{ double t0 = now_s(); ora_process_seqs(&opt, idx, n_processed, n, seqs, op.has_pes0 ? op.pes : 0); t_align += now_s() - t0; }
			n_processed += n;
			for (i = 0; i < n; ++i) {
				if (seqs[i].sam) fputs(seqs[i].sam, stdout);
				free(seqs[i].name); free(seqs[i].seq); free(seqs[i].qual); free(seqs[i].sam);
			}
			free(seqs);
		}
	}
	fprintf(stderr, "[bwa_oracle] aligned %lld reads in %.3f s with %d threads (process_seqs only)\n", (long long)n_processed, t_align, opt.n_threads);
	gzclose(f1.fp); if (f2.fp) gzclose(f2.fp);
	free(f1.line); free(f2.line);
	ora_index_destroy(idx);
	return 0;
}

int main(int argc, char **argv)
{
	if (argc >= 2 && strcmp(argv[1], "mem") == 0) return main_mem(argc - 1, argv + 1);
	if (argc >= 2 && strcmp(argv[1], "stages") == 0) return main_stages(argc - 1, argv + 1);
	fprintf(stderr, "usage: bwa_oracle <mem|stages> ...\n");
	return 1;
}
