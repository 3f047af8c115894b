/* ORACLE (test infrastructure only) -- batch driver.
 * Restates mem_process_seqs (bwamem.c:1215-1244) with its two parallel phases
 * (worker1 bwamem.c:1183, worker2 bwamem.c:1197) and the serial mem_pestat in
 * between.  kt_for's work stealing (kthread.c:49) is replaced by an atomic
 * counter: results do not depend on which thread takes which read.
 */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "ora.h"

typedef struct {
	const ora_opt_t *opt;
	const ora_index_t *idx;
	const ora_pestat_t *pes;
	ora_read_t *seqs;
	ora_reg_v *regs;
	int64_t n_processed;
	long n, next;
	int phase;
} job_t;

static void align_one(job_t *w, long i, ora_aux_t *aux)     /* bwamem.c:1183 worker1 */
{
	if (!(w->opt->flag & ORA_F_PE)) {
		w->regs[i] = ora_align1_core(w->opt, w->idx, w->seqs[i].l_seq, w->seqs[i].seq, aux);
	} else {
		w->regs[i<<1|0] = ora_align1_core(w->opt, w->idx, w->seqs[i<<1|0].l_seq, w->seqs[i<<1|0].seq, aux);
		w->regs[i<<1|1] = ora_align1_core(w->opt, w->idx, w->seqs[i<<1|1].l_seq, w->seqs[i<<1|1].seq, aux);
	}
}

static void finish_one(job_t *w, long i)                    /* bwamem.c:1197 worker2 */
{
	if (!(w->opt->flag & ORA_F_PE)) {
		ora_mark_primary_se(w->opt, w->regs[i].n, w->regs[i].a, w->n_processed + i);
		if (w->opt->flag & ORA_F_PRIMARY5) ora_reorder_primary5(w->opt->T, &w->regs[i]);
		ora_reg2sam(w->opt, w->idx->ref, &w->seqs[i], &w->regs[i], 0, 0);
		free(w->regs[i].a);
	} else {
		ora_sam_pe(w->opt, w->idx->ref, w->pes, (w->n_processed >> 1) + i, &w->seqs[i<<1], &w->regs[i<<1]);
		free(w->regs[i<<1|0].a); free(w->regs[i<<1|1].a);
	}
}

static void *thread_main(void *arg)
{
	job_t *w = (job_t*)arg;
	ora_aux_t *aux = w->phase == 1 ? ora_aux_new() : 0;
	for (;;) {
		long i = __sync_fetch_and_add(&w->next, 1);
		if (i >= w->n) break;
		if (w->phase == 1) align_one(w, i, aux);
		else finish_one(w, i);
	}
	if (aux) ora_aux_free(aux);
	return 0;
}

static void run_phase(job_t *w, int phase, int n_threads)
{
	pthread_t *tid;
	int t;
	w->phase = phase; w->next = 0;
	if (n_threads <= 1) { thread_main(w); return; }
	tid = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
	for (t = 0; t < n_threads; ++t) pthread_create(&tid[t], 0, thread_main, w);
	for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
	free(tid);
}

void ora_process_seqs(const ora_opt_t *opt, const ora_index_t *idx, int64_t n_processed, int n, ora_read_t *seqs, const ora_pestat_t *pes0)
{
	job_t w;
	ora_pestat_t pes[4];
	memset(&w, 0, sizeof w);
	w.opt = opt; w.idx = idx; w.seqs = seqs; w.n_processed = n_processed; w.pes = pes;
	w.regs = (ora_reg_v*)calloc(n ? n : 1, sizeof(ora_reg_v));
	w.n = (opt->flag & ORA_F_PE) ? n >> 1 : n;
	run_phase(&w, 1, opt->n_threads);
	if (opt->flag & ORA_F_PE) {
		if (pes0) memcpy(pes, pes0, sizeof pes);
		else ora_pestat(opt, idx->ref->l_pac, n, w.regs, pes);
	}
	run_phase(&w, 2, opt->n_threads);
	free(w.regs);
}
