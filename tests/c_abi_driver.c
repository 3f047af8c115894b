/* Plain-C caller of the drop-in boundary (no Python, no ctypes): links libbwahip.so and libbwamem_hip.so, builds the
 * reference-shaped bwt_t / bntseq_t / pac through the loader, and drives the three entry points a reference
 * translation unit would use:
 *     bwahip_init (the structs a bwa_idx_load()ed index consists of)  ->  bwahip_align_batch  (== kt_for(worker1))
 *     mem_process_seqs (bwamem.h:69 signature, from libbwamem_hip.so) ->  SAM text on stdout
 * Usage: c_abi_driver [-p] [-a] [-K reads_per_batch] [-R rg_id] <prefix> <reads.fq> [mates.fq]
 *        c_abi_driver -F [-a] [-K bases_per_batch] [-g devices] [-c contexts_per_device] [-t host_threads] <prefix> <reads.fq[.gz]> [mates.fq[.gz]]
 *            bwahip_stream_run, the library's batch driver (superBatchMain's role): one reader, one worker per context
 *            (bwahip_ctx_clone_on: devices; further contexts share a device's index), SAM in input order on stdout
 * Prints the SAM body; on stderr "regs <n_reads> <n_regions> <checksum>" from the bwahip_align_batch pass. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/bwamem_hip.h"   /* mem_process_seqs (libbwamem_hip.so) + bwahip.h (libbwahip.so) */

static char *dupn(const char *s, size_t n) { char *p = (char*)malloc(n + 1); memcpy(p, s, n); p[n] = 0; return p; }

static int read_fastq(const char *fn, bwahip_seq_t **out)
{
	FILE *fp = fopen(fn, "r");
	char *line = 0; size_t cap = 0; ssize_t l;
	int n = 0, m = 0;
	bwahip_seq_t *a = 0;
	if (!fp) { perror(fn); exit(1); }
	while ((l = getline(&line, &cap, fp)) > 0) {
		bwahip_seq_t s;
		size_t k = 1;
		if (line[0] != '@') continue;
		memset(&s, 0, sizeof s);
		while (line[k] && line[k] != ' ' && line[k] != '\t' && line[k] != '\n') ++k;
		if (k > 3 && line[k-2] == '/' && line[k-1] >= '0' && line[k-1] <= '9') k -= 2;   /* bwa.c:73 trim_readno */
		s.name = dupn(line + 1, k - 1);
		if ((l = getline(&line, &cap, fp)) <= 0) break;
		while (l && (line[l-1] == '\n' || line[l-1] == '\r')) --l;
		s.seq = dupn(line, l); s.l_seq = (int)l;
		if (getline(&line, &cap, fp) <= 0) break;
		if ((l = getline(&line, &cap, fp)) <= 0) break;
		while (l && (line[l-1] == '\n' || line[l-1] == '\r')) --l;
		s.qual = dupn(line, l);
		if (n == m) { m = m ? m << 1 : 1024; a = (bwahip_seq_t*)realloc(a, (size_t)m * sizeof *a); }
		s.id = n; a[n++] = s;
	}
	free(line); fclose(fp);
	*out = a;
	return n;
}

int main(int argc, char **argv)
{
	bwahip_opt_t opt;
	bwahip_ctx *loader = 0, *ctx = 0;
	bwahip_seq_t *s1 = 0, *s2 = 0, *seqs;
	int i, n1, n, per = 1 << 30, ai = 1, rc, files_mode = 0, n_dev = 1, ctx_per_dev = 2;
	bwahip_opt_init(&opt);
	for (; ai < argc && argv[ai][0] == '-'; ++ai) {
		if (!strcmp(argv[ai], "-p")) opt.flag |= BWAHIP_F_PE;
		else if (!strcmp(argv[ai], "-F")) files_mode = 1;
		else if (!strcmp(argv[ai], "-a")) opt.flag |= BWAHIP_F_ALL;
		else if (!strcmp(argv[ai], "-g") && ai + 1 < argc) n_dev = atoi(argv[++ai]);
		else if (!strcmp(argv[ai], "-c") && ai + 1 < argc) ctx_per_dev = atoi(argv[++ai]);
		else if (!strcmp(argv[ai], "-K") && ai + 1 < argc) per = atoi(argv[++ai]);
		else if (!strcmp(argv[ai], "-t") && ai + 1 < argc) opt.n_threads = atoi(argv[++ai]);
		else if (!strcmp(argv[ai], "-R") && ai + 1 < argc) bwahip_compat_set_rg_id(argv[++ai]);
		else { fprintf(stderr, "unknown option %s\n", argv[ai]); return 2; }
	}
	if (ai + 2 > argc) { fprintf(stderr, "usage: c_abi_driver [-p] [-a] [-K n] [-t n] [-R id] <prefix> <reads.fq> [mates.fq]\n"); return 2; }
	/* the index as a reference program holds it after bwa_idx_load(): bwt_t, bntseq_t, pac on the host */
	if ((rc = bwahip_init_from_files(argv[ai], 0, &loader))) { fprintf(stderr, "index load failed: %d\n", rc); return 1; }
	if (files_mode) {
		/* the library's batch driver: one reader, ctx_per_dev contexts on each of n_dev devices (more devices than the box has:
		 * further contexts on the devices it has), whole -K batches dealt with their true n_processed, SAM in input order on stdout */
		bwahip_ctx *cx[64];
		bwahip_stream_t st;
		int k, n_cx = 0, have = bwahip_device_count();
		memset(&st, 0, sizeof st);
		st.chunk_bases = per == 1 << 30 ? 0 : per;
		if (n_dev < 1) n_dev = 1;
		if (ctx_per_dev < 1) ctx_per_dev = 1;
		if (n_dev * ctx_per_dev > 64) { fprintf(stderr, "too many contexts\n"); return 2; }
		cx[n_cx++] = loader;
		for (k = 1; k < n_dev * ctx_per_dev; ++k) {
			const int dev = have > 0 ? (k / ctx_per_dev) % have : 0;
			if ((rc = bwahip_ctx_clone_on(loader, dev, &cx[n_cx]))) { fprintf(stderr, "bwahip_ctx_clone_on(%d) failed: %d\n", dev, rc); return 1; }
			++n_cx;
		}
		fflush(stdout);
		rc = bwahip_stream_run(cx, n_cx, &opt, 0, argv[ai + 1], ai + 2 < argc ? argv[ai + 2] : 0, 1, &st);
		for (k = n_cx - 1; k >= 0; --k) bwahip_destroy(cx[k]);
		if (rc) { fprintf(stderr, "bwahip_stream_run failed: %d\n", rc); return 1; }
		fprintf(stderr, "files %lld reads %lld batches %d contexts %.3f s\n", (long long)st.n_reads, (long long)st.n_batches, n_cx, st.seconds);
		return 0;
	}
	n1 = read_fastq(argv[ai + 1], &s1);
	if (ai + 2 < argc) {
		int n2 = read_fastq(argv[ai + 2], &s2);
		if (n2 != n1) { fprintf(stderr, "mate files differ in length\n"); return 1; }
		opt.flag |= BWAHIP_F_PE;
		seqs = (bwahip_seq_t*)malloc((size_t)2 * n1 * sizeof *seqs);
		for (i = 0; i < n1; ++i) { seqs[2*i] = s1[i]; seqs[2*i+1] = s2[i]; }
		n = 2 * n1;
	} else { seqs = s1; n = n1; }

	/* 1. bwahip_init on caller-owned structs + bwahip_align_batch on copies of the reads (it converts seq in place) */
	if ((rc = bwahip_init(bwahip_bwt(loader), bwahip_bns(loader), bwahip_pac(loader), 0, &ctx))) { fprintf(stderr, "bwahip_init failed: %d\n", rc); return 1; }
	{
		bwahip_seq_t *cp = (bwahip_seq_t*)malloc((size_t)n * sizeof *cp);
		bwahip_alnreg_v *regs = (bwahip_alnreg_v*)calloc(n, sizeof *regs);
		long long n_regs = 0; unsigned long long sum = 0;
		for (i = 0; i < n; ++i) { cp[i] = seqs[i]; cp[i].seq = dupn(seqs[i].seq, seqs[i].l_seq); }
		if ((rc = bwahip_align_batch(ctx, &opt, n, cp, regs))) { fprintf(stderr, "bwahip_align_batch failed: %d\n", rc); return 1; }
		for (i = 0; i < n; ++i) {
			int k;
			for (k = 0; k < regs[i].n; ++k) {
				const bwahip_alnreg_t *p = &regs[i].a[k];
				sum = sum * 1000003ULL + (unsigned long long)p->rb * 31 + (unsigned long long)p->re * 17 + (unsigned)p->qb * 7 + (unsigned)p->qe * 5 + (unsigned)p->score;
			}
			n_regs += regs[i].n;
			free(regs[i].a); free(cp[i].seq);
		}
		fprintf(stderr, "regs %d %lld %llu\n", n, n_regs, sum);
		free(cp); free(regs);
	}
	bwahip_destroy(ctx);

	/* 2. mem_process_seqs with the reference's signature, in batches like the reference's pipeline step */
	{
		int64_t n_processed = 0;
		if (opt.flag & BWAHIP_F_PE) per &= ~1;
		while (n_processed < n) {
			int nb = n - n_processed < per ? (int)(n - n_processed) : per;
			mem_process_seqs(&opt, bwahip_bwt(loader), bwahip_bns(loader), bwahip_pac(loader), n_processed, nb, seqs + n_processed, 0);
			for (i = 0; i < nb; ++i) { fputs(seqs[n_processed + i].sam, stdout); free(seqs[n_processed + i].sam); }
			n_processed += nb;
		}
	}
	bwahip_destroy(loader);
	return 0;
}
