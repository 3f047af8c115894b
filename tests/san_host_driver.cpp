// TEST INFRASTRUCTURE (tests/test_sanitizers.py): the product's host-side translation units -- csrc/index_io.cpp (index loader)
// and csrc/host_final.cpp (mem_process_seqs' worker2 on host threads: the gpu_final = 0 / gpu_pair = 0 path) -- compiled with
// -fsanitize=address,undefined and driven without a GPU.  The one thing those files take from the GPU, bwahip_align_batch
// (kt_for(worker1), bwamem.c:1232), is supplied here by the CPU oracle (oracle/liboracle: ora_align1_core), so that every line of
// the host code runs on the golden reads under the sanitizers and its SAM can be compared with the reference's.
//   san_host_driver [-p] [-a] [-t N] <index prefix> <reads.fq> [mates.fq]   > SAM records (no header)
#include "../bwa-mem-gpu_amd/csrc/bwahip_internal.h"
extern "C" {
#include "../oracle/ora.h"
}
#include <limits.h>
extern "C" int bwahip_process_seqs_host(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0);   // host_final.cpp

struct bwahip_ctx { HostIndex host; ora_index_t *oidx; };

static_assert(sizeof(ora_opt_t) == sizeof(bwahip_opt_t), "both mirror mem_opt_t field for field");

const bwahip_bns_t *bwahip_bns(const bwahip_ctx *c) { return &c->host.bns; }
const uint8_t *bwahip_pac(const bwahip_ctx *c) { return c->host.pac; }
const char *bwahip_ctx_rg_id(const bwahip_ctx *) { return ""; }

// kt_for(worker1): mem_align1_core per read (converts seq to 0..4 codes in place, bwamem.c:1067), results malloc()ed per read
int bwahip_align_batch(bwahip_ctx *c, const bwahip_opt_t *opt, int n, bwahip_seq_t *seqs, bwahip_alnreg_v *out)
{
	ora_opt_t o;
	memcpy(&o, opt, sizeof o);
	ora_aux_t *aux = ora_aux_new();
	for (int i = 0; i < n; ++i) {
		ora_reg_v v = ora_align1_core(&o, c->oidx, seqs[i].l_seq, seqs[i].seq, aux);
		out[i].n = out[i].m = v.n;
		out[i].a = (bwahip_alnreg_t*)calloc(v.n ? v.n : 1, sizeof(bwahip_alnreg_t));
		for (int k = 0; k < v.n; ++k) {
			const ora_reg_t &s = v.a[k];
			bwahip_alnreg_t &d = out[i].a[k];
			d.rb = s.rb; d.re = s.re; d.hash = s.hash; d.frac_rep = s.frac_rep; d.qb = s.qb; d.qe = s.qe; d.rid = s.rid; d.score = s.score;
			d.truesc = s.truesc; d.sub = s.sub; d.alt_sc = s.alt_sc; d.csub = s.csub; d.sub_n = s.sub_n; d.w = s.w; d.seedcov = s.seedcov;
			d.secondary = s.secondary; d.secondary_all = s.secondary_all; d.seedlen0 = s.seedlen0; d.n_comp = s.n_comp; d.is_alt = s.is_alt;
		}
		free(v.a);
	}
	ora_aux_free(aux);
	return 0;
}

int main(int argc, char **argv)
{
	bwahip_opt_t opt;
	ora_opt_t oo;
	ora_opt_init(&oo);
	memcpy(&opt, &oo, sizeof opt);
	int ai = 1;
	long long dump_chunk = 0;                                  // -d CHUNK: only read the input and print the batches (format of `bwaref readfq`)
	opt.n_threads = 3;
	for (; ai < argc && argv[ai][0] == '-'; ++ai) {
		if (!strcmp(argv[ai], "-p")) opt.flag |= BWAHIP_F_PE;
		else if (!strcmp(argv[ai], "-a")) opt.flag |= BWAHIP_F_ALL;
		else if (!strcmp(argv[ai], "-d") && ai + 1 < argc) dump_chunk = atoll(argv[++ai]);
		else if (!strcmp(argv[ai], "-t") && ai + 1 < argc) opt.n_threads = atoi(argv[++ai]);
		else { fprintf(stderr, "unknown option %s\n", argv[ai]); return 2; }
	}
	if (dump_chunk > 0 && argc - ai >= 1) {
		bwahip_fastq *rd = nullptr;
		if (bwahip_fastq_open(argv[ai], argc - ai > 1 ? argv[ai + 1] : nullptr, &rd)) return 1;
		for (;;) {
			bwahip_seq_t *sp = nullptr; int n = 0;
			if (bwahip_fastq_next(rd, dump_chunk, 1, &sp, &n)) return 1;
			if (n == 0) break;
			printf("#batch %d\n", n);
			for (int i = 0; i < n; ++i) printf("%s\t%s\t%s\t%s\n", sp[i].name, sp[i].comment ? sp[i].comment : "*", sp[i].seq, sp[i].qual ? sp[i].qual : "*");
		}
		bwahip_fastq_close(rd);
		return 0;
	}
	if (argc - ai < 2) { fprintf(stderr, "usage: san_host_driver [-p] [-a] [-t N] <prefix> <reads.fq> [mates.fq]\n"); return 2; }
	bwahip_ctx ctx;
	if (bwahip_load_index_files(argv[ai], &ctx.host)) { fprintf(stderr, "index load failed\n"); return 1; }
	ctx.oidx = ora_index_load(argv[ai]);
	if (!ctx.oidx) { fprintf(stderr, "oracle index load failed\n"); return 1; }
	// the product's own FASTQ reader (csrc/fastq_reader.cpp) supplies the batch: it runs under the sanitizers too
	const int n_files = argc - ai - 1;
	bwahip_fastq *rd = nullptr;
	if (bwahip_fastq_open(argv[ai + 1], n_files == 2 ? argv[ai + 2] : nullptr, &rd)) { fprintf(stderr, "cannot read the input\n"); return 1; }
	if (n_files == 2) opt.flag |= BWAHIP_F_PE;
	bwahip_seq_t *sp = nullptr; int n = 0;
	if (bwahip_fastq_next(rd, INT64_MAX / 2, 0, &sp, &n)) return 1;
	std::vector<bwahip_seq_t> seqs(sp, sp + n);
	// two batches with their true n_processed (hash_64 tie-breaks, bwamem.c:1204,1210), as the reference cuts a long input
	const int half = (opt.flag & BWAHIP_F_PE) ? n : (n / 2) & ~1;      // PE: one batch (mem_pestat is per batch, and the golden run had one)
	int rc = bwahip_process_seqs_host(&ctx, &opt, 0, half, seqs.data(), nullptr);
	if (!rc && n > half) rc = bwahip_process_seqs_host(&ctx, &opt, half, n - half, seqs.data() + half, nullptr);
	if (rc) { fprintf(stderr, "bwahip_process_seqs_host: %d\n", rc); return 1; }
	char *sam = nullptr; int64_t len = 0;
	if (bwahip_seqs_take_sam(seqs.data(), n, &sam, &len)) return 1;
	fwrite(sam, 1, (size_t)len, stdout);
	free(sam);
	bwahip_fastq_close(rd);
	bwahip_free_host_index(&ctx.host);
	ora_index_destroy(ctx.oidx);
	return 0;
}
