/* bwahip.h -- C ABI of the MI355X-native BWA-MEM hot path.
 *
 * This is the drop-in boundary: a thin `extern "C"` launcher that replaces the
 * reference's CUDA seam (cuda/bwamem_GPU.cuh:13 mem_align_GPU,
 * cuda/streams.cuh:21-63 newProcess/newTransfer/..., cuda/bwt_CUDA.cuh,
 * cuda/ksw_CUDA.cuh) behind the unchanged host surface
 *     mem_process_seqs()   bwamem.h:69 / bwamem.c:1215
 *     mem_align1_core()    bwamem.c:1061   (= the per-read work of worker1, bwamem.c:1183)
 * Plain pointers and sizes only; no C++/torch types.  Every struct below is a
 * layout mirror of the reference's own struct (cited), so a reference
 * translation unit can pass its objects straight through.
 * All functions return 0 on success and a negative BWAHIP_E* code on failure;
 * nothing here ever falls back to a CPU implementation of the hot path.
 */
#ifndef BWAHIP_H
#define BWAHIP_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- layout mirrors of reference types --------------------------------- */

typedef struct {                 /* bwt_t, bwt.h:48-60 */
	uint64_t primary;
	uint64_t L2[5];
	uint64_t seq_len;
	uint64_t bwt_size;           /* number of uint32 words in `bwt` */
	uint32_t *bwt;               /* Occ-interleaved BWT: per 128 bases 4 x u64 counts + 8 x u32 (bwtindex.c:150) */
	uint32_t cnt_table[256];
	int sa_intv;
	uint64_t n_sa;
	uint64_t *sa;
} bwahip_bwt_t;

typedef struct {                 /* bntann1_t, bntseq.h:41-48 */
	int64_t offset;
	int32_t len;
	int32_t n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} bwahip_ann_t;

typedef struct {                 /* bntamb1_t, bntseq.h:50-54 */
	int64_t offset;
	int32_t len;
	char amb;
} bwahip_amb_t;

typedef struct {                 /* bntseq_t, bntseq.h:56-64 */
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bwahip_ann_t *anns;
	int32_t n_holes;
	bwahip_amb_t *ambs;
	FILE *fp_pac;
} bwahip_bns_t;

typedef struct {                 /* bseq1_t, bwa.h:58-63 (fork layout: l_name/l_comment/l_qual appended) */
	int l_seq, id;
	char *name, *comment, *seq, *qual, *sam;
	int8_t l_name, l_comment;
	int16_t l_qual;
} bwahip_seq_t;

typedef struct {                 /* mem_opt_t, bwa.h:86-118 (168 bytes, mat at offset 136) */
	uint64_t max_mem_intv;
	int a, b;
	int o_del, e_del;
	int o_ins, e_ins;
	int pen_unpaired;
	int pen_clip5, pen_clip3;
	int w;
	int zdrop;
	int T;
	int flag;
	int min_seed_len;
	int min_chain_weight;
	int max_chain_extend;
	float split_factor;
	int split_width;
	int max_occ;
	int max_chain_gap;
	int n_threads;
	int chunk_size;
	float mask_level;
	float drop_ratio;
	float XA_drop_ratio;
	float mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins;
	int max_matesw;
	int max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} bwahip_opt_t;

#define BWAHIP_F_PE        0x2      /* MEM_F_* of bwa.h:70-81 */
#define BWAHIP_F_NOPAIRING 0x4
#define BWAHIP_F_ALL       0x8
#define BWAHIP_F_NO_MULTI  0x10
#define BWAHIP_F_NO_RESCUE 0x20
#define BWAHIP_F_REF_HDR   0x100
#define BWAHIP_F_SOFTCLIP  0x200
#define BWAHIP_F_SMARTPE   0x400
#define BWAHIP_F_PRIMARY5  0x800
#define BWAHIP_F_KEEP_SUPP_MAPQ 0x1000
#define BWAHIP_F_XB        0x2000

typedef struct {                 /* mem_alnreg_t, bwa.h:145-163 (88 bytes) */
	int64_t rb, re;
	uint64_t hash;
	float frac_rep;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp:30, is_alt:2;
} bwahip_alnreg_t;

typedef struct { int n, m; bwahip_alnreg_t *a; } bwahip_alnreg_v;   /* mem_alnreg_v, bwa.h:165 */

typedef struct {                 /* mem_pestat_t, bwa.h:167-171 */
	int low, high;
	int failed;
	double avg, std;
} bwahip_pestat_t;

typedef struct { uint64_t x[3], info; } bwahip_intv_t;              /* bwtintv_t, bwt.h:62-64 */

/* ---- error codes -------------------------------------------------------- */
#define BWAHIP_OK          0
#define BWAHIP_EINVAL     -1   /* bad argument */
#define BWAHIP_ENODEV     -2   /* no usable HIP device / HIP runtime error (message on stderr) */
#define BWAHIP_ENOMEM     -3   /* device or host allocation failed */
#define BWAHIP_EIO        -4   /* index files unreadable / inconsistent */
#define BWAHIP_ECAPACITY  -5   /* a read exceeds the compiled limits (length > BWAHIP_MAX_READ_LEN) */
#define BWAHIP_EINTERNAL  -6   /* a kernel reported an inconsistency (never expected) */

/* Longest read the kernels accept (LDS sizing of the per-read kernels).  mem_flt_chained_seeds (bwamem.c:605) runs on the GPU
 * (k_seed_sw): with the default -W 0 it is active only from ~730 bp, but a caller's -W (opt->min_chain_weight) switches it on
 * for every read of at least 22*W bases. */
#define BWAHIP_MAX_READ_LEN 700

typedef struct bwahip_ctx bwahip_ctx;

/* ---- reading FASTA/FASTQ (plain, gzip or bgzip) into batches: bseq_read (bwa.c:191) / kseq_read (kseq.h:176) -------------
 * A parallel pipeline per input file reads ahead of the caller (csrc/fastq_reader.cpp): plain files are mmap()ed and parsed in
 * chunks by n_threads workers, gzip streams are inflated on their own thread with the parse workers behind it, BGZF (bgzip)
 * members are inflated by the workers in parallel; anything that is not plain four-line FASTQ goes through an exact, sequential
 * restatement of kseq_read.  path2 != NULL: the mates' file, batches come interleaved (read i of file 1, read i of file 2);
 * "-" = stdin.  n_threads <= 0: BWAHIP_READER_THREADS or half of the host's cores (at most 8). */
typedef struct bwahip_fastq bwahip_fastq;
typedef struct bwahip_fastq_batch bwahip_fastq_batch;
int  bwahip_fastq_open(const char *path1, const char *path2, bwahip_fastq **out);
int  bwahip_fastq_open_mt(const char *path1, const char *path2, int n_threads, bwahip_fastq **out);
/* Next batch: reads until it holds at least chunk_bases bases and an even number of reads (bwa.c:216; `bwa mem -K`).  *n = 0 at
 * the end of the input.  name/comment/seq/qual of (*seqs)[i] point into the reader's memory and stay valid until the next call
 * or bwahip_fastq_close -- unlike bseq_read's they are NOT the caller's to free; seqs[i].sam (set by bwahip_process_seqs) is.
 * keep_comments = 0 drops FASTQ comments (stock behaviour without -C).  Names lose a trailing "/[0-9]" (bwa.c:73).
 * A damaged input (corrupt or truncated gzip data) returns BWAHIP_EIO -- the reference's err_gzread (utils.c:142) is fatal --
 * never a silently shortened batch. */
int  bwahip_fastq_next(bwahip_fastq *r, int64_t chunk_bases, int keep_comments, bwahip_seq_t **seqs, int *n);
/* The same batch as an owned object: it stays valid (with its strings) until bwahip_fastq_batch_release, independently of
 * later batches and of the reader, so that several batches can be in flight on several contexts.  *batch = NULL and *n = 0 at
 * the end of the input.  seqs may be NULL (bwahip_fastq_batch_seqs returns the array later). */
int  bwahip_fastq_next_batch(bwahip_fastq *r, int64_t chunk_bases, int keep_comments, bwahip_fastq_batch **batch, bwahip_seq_t **seqs, int *n);
bwahip_seq_t *bwahip_fastq_batch_seqs(bwahip_fastq_batch *b, int *n);
void bwahip_fastq_batch_release(bwahip_fastq_batch *b);
void bwahip_fastq_close(bwahip_fastq *r);

/* ---- lifetime ------------------------------------------------------------
 * bwahip_init replaces newProcess()/transferIndex() (cuda/streams.cu:8,164): it copies the three
 * index arrays (bwt, sa, pac) and the contig table into HBM of HIP device `device` and builds the
 * launch workspaces.  The context keeps its own host copy of the contig table (names included) and of the
 * packed reference; the caller's arrays are not referenced after it returns. */
int  bwahip_init(const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac, int device, bwahip_ctx **out);
/* Same, but bwt_dev->bwt, bwt_dev->sa and pac_dev already point into HBM of `device` (e.g. filled by an RCCL
 * broadcast from the rank that loaded the index); the device arrays stay owned by the caller and must outlive the
 * ctx.  bns is a host struct; the packed reference is read back once (l_pac/4+1 bytes) for host-side finalisation. */
int  bwahip_init_device(const bwahip_bwt_t *bwt_dev, const bwahip_bns_t *bns, const uint8_t *pac_dev, int device, bwahip_ctx **out);
/* One process per GPU on one node: rank 0 reads the index files, every rank receives the three index arrays and the contig
 * table over RCCL (xGMI) into its own HBM and gets a context on them -- what transferIndex() (cuda/streams.cu:8) does for
 * one GPU.  id128: a 128-byte ncclUniqueId made by bwahip_rccl_unique_id() on one rank and handed to the others by the
 * caller's own means (MPI, a file, torch.distributed ...).  prefix is read on rank 0 only.  Collective: every rank calls it. */
int  bwahip_rccl_unique_id(void *id128);
int  bwahip_init_rccl(const char *prefix, int rank, int world, const void *id128, int device, bwahip_ctx **out);
/* A further context on the same GPU sharing src's index arrays in HBM (no device copy; src must outlive it).  Contexts are
 * independent otherwise (own streams and batch buffers): two of them, each driven by its own host thread and taking batches in
 * turn, overlap one batch's latency-bound kernels with the other's throughput-bound ones. */
int  bwahip_ctx_clone(bwahip_ctx *src, bwahip_ctx **out);
/* A context on another GPU of the node: the three index arrays are copied device to device (hipMemcpyPeer: over xGMI where the
 * GPUs are linked, no second pass over the files or the host copy) into HBM the new context owns; it is independent of src
 * afterwards.  device == src's device: the same as bwahip_ctx_clone. */
int  bwahip_ctx_clone_on(bwahip_ctx *src, int device, bwahip_ctx **out);
int  bwahip_device_count(void);              /* HIP devices visible to the process (0 when there is none) */
/* Convenience: read a stock `bwa index` file set <prefix>.{bwt,sa,pac,ann,amb[,alt]} (bwa.c:402 bwa_idx_load) and init. */
int  bwahip_init_from_files(const char *prefix, int device, bwahip_ctx **out);
void bwahip_destroy(bwahip_ctx *ctx);
/* Host copies of the loaded index (valid until destroy); lets a caller that used
 * bwahip_init_from_files get at contig names etc. without loading the index twice. */
const bwahip_bns_t *bwahip_bns(const bwahip_ctx *ctx);
const bwahip_bwt_t *bwahip_bwt(const bwahip_ctx *ctx);
const uint8_t      *bwahip_pac(const bwahip_ctx *ctx);
/* Read-group id printed as RG:Z:<id> on every record (the reference's global bwa_rg_id, bwa.c:44, set by -R); NULL or
 * "" = none. */
int bwahip_ctx_set_rg_id(bwahip_ctx *ctx, const char *id);
const char *bwahip_ctx_rg_id(const bwahip_ctx *ctx);
void bwahip_opt_init(bwahip_opt_t *opt);     /* mem_opt_init defaults, bwamem.c:74 */
void bwahip_opt_fill_scmat(bwahip_opt_t *opt);   /* bwa_fill_scmat(opt->a, opt->b, opt->mat), bwa.c:249: call after changing a or b */

/* ---- the hot path ---------------------------------------------------------
 * bwahip_align_batch == kt_for(worker1) of mem_process_seqs (bwamem.c:1232): for every read i it
 * produces exactly the mem_alnreg_v that mem_align1_core (bwamem.c:1061) returns, computed on the
 * GPU.  seqs[i].seq is converted in place to 0..4 codes as the reference does (bwamem.c:1067).
 * regs_out[i].a is malloc()ed for the caller (free() it), regs_out[i].n == regs_out[i].m. */
int bwahip_align_batch(bwahip_ctx *ctx, const bwahip_opt_t *opt, int n, bwahip_seq_t *seqs, bwahip_alnreg_v *regs_out);

/* bwahip_process_seqs == mem_process_seqs (bwamem.h:69): hot path on the GPU, then the per-read
 * finalisation (mark primary, mapQ, CIGAR/NM/MD, SAM text; PE: insert-size stats, mate rescue,
 * pairing) with opt->n_threads host threads.  seqs[i].sam is malloc()ed, NUL terminated. */
int bwahip_process_seqs(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0);
/* The same work with the batch's SAM in one piece, for a caller whose output step is a single fwrite: *sam = NUL-terminated text
 * of the whole batch in read order (*sam_len bytes), *off (may be NULL) = n + 1 offsets with read i's records at sam[off[i]..off[i+1]).
 * Both point into the context; seqs[i].sam is left NULL (no malloc per read).  *off stays valid until the next call on the context,
 * *sam until the next-but-one (the context alternates between two pinned buffers, so that a writer thread can still be busy with
 * batch k while batch k + 1 is processed). */
int  bwahip_process_seqs_text(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs,
                              const bwahip_pestat_t *pes0, const char **sam, int64_t *sam_len, const int64_t **off);

/* ---- the batch driver: FASTQ files in -> SAM text out -------------------------------------------------------------------
 * What superBatchMain(ktp_aux_t*) (cuda/superbatch_process.h:35, superbatch_process.cpp:133: read || process, double buffered,
 * one GPU) and process()/kt_pipeline (fastmap.c:46,307: bseq_read -> mem_process_seqs -> fputs) are in the reference, for any
 * number of contexts: one reader (bwahip_fastq_*), one host thread per context taking whole batches with their true
 * n_processed, one writer emitting the SAM text in input order to out_fd (< 0: the text is produced and dropped).  ctxs: n_ctx
 * contexts -- on different devices (bwahip_ctx_clone_on), or several on one device (bwahip_ctx_clone; two per device overlap one
 * batch's serial tails with the other's kernels).  fq2 != NULL: paired-end (MEM_F_PE is set).  opt->n_threads is the host-thread
 * budget of the run for batch staging (divided among the contexts).  Fill the first four fields of *st (0 = defaults); the rest
 * is written on return.  The SAM header is the caller's (bwa_print_sam_hdr, bwa.c:520).  Returns the first error of any stage. */
typedef struct {
	int64_t chunk_bases;      /* in: bases per batch, bwa mem -K (actual_chunk_size, fastmap.c:304); <= 0: opt->chunk_size * opt->n_threads */
	int64_t max_reads;        /* in: > 0: stop at the first batch boundary at or after this many reads */
	int keep_comments;        /* in: -C */
	int reader_threads;       /* in: parse threads of the reader; <= 0: default */
	int64_t n_reads, n_batches, sam_bytes;   /* out */
	double seconds;           /* out: opening the files -> last SAM byte written */
	double reader_wait_s;     /* out: summed over the workers: time spent waiting for the reader */
	double gpu_busy_s;        /* out: summed over the workers: time inside bwahip_process_seqs_text */
	double write_s;           /* out: time the writer spent in write() */
} bwahip_stream_t;
int bwahip_stream_run(bwahip_ctx *const *ctxs, int n_ctx, const bwahip_opt_t *opt, const bwahip_pestat_t *pes0,
                      const char *fq1, const char *fq2, int out_fd, bwahip_stream_t *st);

/* Insert-size statistics (mem_pestat_t[4]: FF, FR, RF, RR; bwamem_pair.c:72) and mate-rescue counters ([0] local alignments
 * run, [1] regions added, [2] most alignments of one pair, [3] pairs that needed any; bwamem_pair.c:137) of the last
 * paired-end batch finalised on the GPU.  Either pointer may be NULL; counters4 receives 4 values. */
int bwahip_last_pe_stats(bwahip_ctx *ctx, bwahip_pestat_t *pes4, uint64_t *counters4);

/* Concatenate seqs[0..n).sam into one malloc()ed buffer (read order; *out_len bytes + a NUL) and free the per-read strings. */
int bwahip_seqs_take_sam(bwahip_seq_t *seqs, int n, char **out, int64_t *out_len);

/* ---- stage-level entry points (parity tests and bench) --------------------
 * Reads are given packed: `seq` holds the concatenated 0..4 codes, read i is seq[off[i] .. off[i+1]).
 * Results come back as "i64 record" streams in malloc()ed buffers (*out, *out_len int64 words) in the
 * same layout the oracle's stage dump uses (oracle/ref_driver.c): see bwahip_stage_* tags. */
#define BWAHIP_STAGE_INTV      1   /* intervals after mem_collect_intv (bwamem.c:137) */
#define BWAHIP_STAGE_CHAIN     2   /* chains after mem_chain (bwamem.c:258), B-tree order */
#define BWAHIP_STAGE_CHAIN_FLT 3   /* chains after mem_chain_flt (bwamem.c:334) */
#define BWAHIP_STAGE_REGS_PRE  5   /* regions after all mem_chain2aln calls (bwamem.c:1079) */
#define BWAHIP_STAGE_REGS      4   /* regions returned by mem_align1_core */
int bwahip_run_stages(bwahip_ctx *ctx, const bwahip_opt_t *opt, int n, const uint8_t *seq, const int64_t *off,
                      int stage_mask, int64_t **out, int64_t *out_len);

/* Device-resident batch for benchmarking: upload once, run the whole hot path (seq codes in HBM ->
 * alignment regions in HBM) any number of times.  kernel_ms (may be NULL) receives the per-kernel
 * durations of the last run measured with HIP events on the launch stream, in launch order
 * (see bwahip_kernel_name). */
int bwahip_batch_upload(bwahip_ctx *ctx, int n, const uint8_t *seq, const int64_t *off);
/* The same for reads that already sit in HBM of the context's device (codes 0..4 concatenated, off_dev[0] == 0): nothing
 * is copied, the buffers stay the caller's and must stay valid until the next upload / attach / destroy. */
int bwahip_batch_attach(bwahip_ctx *ctx, int n, const uint8_t *seq_dev, const int64_t *off_dev, int max_len, int64_t total_bases);
int bwahip_batch_run(bwahip_ctx *ctx, const bwahip_opt_t *opt, float *kernel_ms, int n_kernel_ms);
int bwahip_batch_download(bwahip_ctx *ctx, bwahip_alnreg_v *regs_out);       /* regs of the last run */
/* The whole of mem_process_seqs on a device-resident batch: attach the text the SAM stage prints (qualities: read r at
 * qual_dev + qual_off_dev[r], or qual_off_dev[r] < 0 / qual_dev NULL for none; NUL-terminated names at names_dev +
 * name_off_dev[r], buffer padded by 64 bytes), run hot path + finalisation + SAM text on the GPU (SE, or PE when
 * opt->flag has MEM_F_PE; n_processed / pes0 as in mem_process_seqs), then fetch the text.  kernel_ms: as
 * bwahip_batch_run; entries 11..15 are the finalisation stages (see bwahip_kernel_name). */
int bwahip_batch_attach_text(bwahip_ctx *ctx, const uint8_t *qual_dev, const int64_t *qual_off_dev, const uint8_t *names_dev, const int64_t *name_off_dev);
int bwahip_batch_run_sam(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, const bwahip_pestat_t *pes0, float *kernel_ms, int n_kernel_ms);
int bwahip_batch_sam(bwahip_ctx *ctx, char **out, int64_t *out_len, int64_t *off);
int bwahip_n_kernels(void);
const char *bwahip_kernel_name(int i);
/* Algorithmic work counters of the last bwahip_batch_run, counted on the device by the kernels
 * themselves: [0] bwt_extend calls, [1] Occ blocks touched by them,
 * [2] bwt_sa calls, [3] LF steps, [4] intervals written, [5] seeds, [6] DP cells, [7] most bwt_extend
 * calls of one read; [8..15] per-phase maxima over reads (10 ns ticks) and the largest seed / chain
 * counts of one read; [16,17] Occ blocks / intervals of k_smem_heavy, [24,25] of k_smem3 (both are
 * included in [1] and [4]; bench.py subtracts them to attribute bytes to k_smem); [19,20] DP rows
 * with one / several columns per lane; [21..23] dedup phase maxima.  n <= 32. */
int bwahip_batch_counters(bwahip_ctx *ctx, uint64_t *counters, int n);

/* Known-answer helpers used by the parity tests: device Occ/extend/SA on arrays of inputs. */
int bwahip_kat_occ4(bwahip_ctx *ctx, int n, const uint64_t *k, uint64_t *cnt4_out);
/* What the index occupies in HBM: interval of the SA table the kernels read (1 = every BWT row; the index files hold every 32nd, bwt.c:86
 * walks the rest), longest string of the interval table (0: none), and the bytes of {BWT with Occ counts, SA table, packed reference,
 * interval table}.  Tunables: BWAHIP_SA_INTV, BWAHIP_KMER_K. */
int bwahip_index_footprint(bwahip_ctx *ctx, int *sa_intv, int *kmer_k, uint64_t *bytes4);
int bwahip_kat_sa(bwahip_ctx *ctx, int n, const uint64_t *k, uint64_t *sa_out);
/* The interval table of the BWT search (strings of up to *k_out bases): every entry of every length against bwt_extend (bwt.c:262) run
 * FORWARD from the entry of the string without its last base (the table itself is filled by backward extensions); *bad_out = mismatches. */
int bwahip_kat_kmer_table(bwahip_ctx *ctx, int *k_out, uint64_t *bad_out);
int bwahip_kat_extend(bwahip_ctx *ctx, int n, const uint64_t *ik3, const int *is_back, uint64_t *ok12_out);
int bwahip_kat_ksw_extend(bwahip_ctx *ctx, int n, const int *params /*n x 10*/, const uint8_t *q, const int64_t *qoff,
                          const uint8_t *t, const int64_t *toff, int *out6 /*n x 6*/);

/* The region-list sorts (ks_introsort over mem_ars2 / mem_ars keys, bwamem.c:398-402, ksort.h:176) as the kernels run them: the whole
 * wavefront's exact form (csrc/isort_dev.h) and the one-lane restatement of ksort.h on the same n keys {k64, score, qb} (mode 0: by k64;
 * mode 1: score descending, k64, qb).  idx_par / idx_seq: the two permutations; status2[0] = 1 when the parallel form ran to the end
 * (0: the introsort's depth limit -- it hands over to the one-lane form, idx_par is the identity), status2[1] != 0: internal error. */
int bwahip_kat_introsort(bwahip_ctx *ctx, int n, int mode, const int64_t *k64, const int *score, const int *qb, int *idx_par, int *idx_seq, int *status2);

/* ksw_align2 (ksw.c:343) on the device, byte or word kernel as xtra's KSW_XBYTE says.  params: n x 8 ints
 * (qlen, tlen, xtra, o_del, e_del, o_ins, e_ins, 0); mat25 NULL = the default 1/-4 matrix; out7: n x 7
 * (score, te, qe, score2, te2, tb, qb). */
int bwahip_kat_ksw_align(bwahip_ctx *ctx, int n, const int *params, const int8_t *mat25, const uint8_t *q, const int64_t *qoff,
                         const uint8_t *t, const int64_t *toff, int *out7);

/* Tuning knobs of the heavy-read hand-off kernels (tests force each one onto ordinary reads): keys intv_cap,
 * smem_lanes, heavy_mult, chain_big_min, rank_sort_min, spec_min_chains, ext_lds_window, verbose.  The same knobs are read from the
 * environment (BWAHIP_<KEY>) once, when the context is created. */
int bwahip_ctx_tune(bwahip_ctx *ctx, const char *key, int value);

const char *bwahip_version(void);

#ifdef __cplusplus
}
#endif
#endif
