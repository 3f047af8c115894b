// Internal definitions shared by the HIP translation units of libbwahip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include "../../include/bwahip.h"

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	fprintf(stderr, "[bwahip] %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, hipGetErrorString(e_)); \
	return BWAHIP_ENODEV; } } while (0)

// Contig table entry in HBM (subset of bntann1_t that the kernels need; bntseq.h:41).
struct DevAnn { int64_t offset; int32_t len; int32_t is_alt; };

// Everything a kernel needs to know about the index; passed by value as a kernel argument.
struct DevIndex {
	const uint4 *bwt;        // Occ-interleaved BWT viewed as 64-byte blocks = 4 x uint4 (bwt.h:74-75), the bases re-laid as bit planes (fmi_dev.h)
	const uint64_t *sa;      // sampled SA, sa[0] = -1 (bwt.c:83)
	const uint8_t *pac;      // forward strand, 4 bases/byte, MSB first (bntseq.c:229)
	const DevAnn *anns;
	uint64_t primary, L2[5], seq_len, n_sa;
	int64_t l_pac;
	int sa_intv, sa_shift, n_seqs;
	// the bi-intervals of all strings of 1..kmer_k bases (k_smem.hip, "interval table"); kmer_k = 0: none
	const uint4 *kmer; int kmer_k;
};
// entry of the string s_0 s_1 .. s_{L-1} (s_t at bits 2t of `code`): kmer[kmer_off(L) + code]; levels follow each other, 4^L entries each
__host__ __device__ inline uint64_t kmer_off(int L) { return (0x5555555555555555ull & ((1ull << 2 * L) - 1)) - 1; }

// Options the kernels read (a flat copy of the fields of mem_opt_t that the hot path uses).
struct DevOpt {
	int a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop;
	int min_seed_len, split_width, max_occ, max_chain_gap, max_mem_intv, split_len;
	int min_chain_weight, max_chain_extend;
	float mask_level, drop_ratio, mask_level_redun;
	int8_t mat[25];
	// finalisation (mem_mark_primary_se .. mem_aln2sam) and pairing
	int T, flag, pen_unpaired, max_ins, max_matesw, max_XA_hits, max_XA_hits_alt, mapQ_coef_fac;
	float XA_drop_ratio, mapQ_coef_len;
};

// Work counters kept in HBM, bumped once per wavefront at kernel exit (bwahip_batch_counters).
enum { CNT_EXTEND = 0, CNT_BLOCKS, CNT_SA, CNT_LF, CNT_INTV, CNT_SEEDS, CNT_CELLS, CNT_MAX_EXT /* most bwt_extend calls of one read */,
       CNT_HEAVY_BLOCKS = 16, CNT_HEAVY_INTV, CNT_HEAVY_READS, CNT_ROWS1 /* DP rows, 1 column per lane */, CNT_ROWSN /* DP rows, CPL columns per lane */,
       CNT_P3_BLOCKS = 24, CNT_P3_INTV, CNT_P3_JUMPED /* pass-3 extensions the interval table stood in for */, CNT_N = 32 };
// The counters are kept in CNT_SLOTS copies (rows of CNT_N); a wavefront updates the row picked by its position in the
// launch and the host folds the rows (sum, or max for the *_max entries).  One shared row made every wavefront's
// end-of-work atomics queue up on the same L2 line: with a million wavefronts that alone cost tens of milliseconds.
constexpr int CNT_SLOTS = 256;
#define BWAHIP_MISC_BYTES ((size_t)CNT_SLOTS * CNT_N * 8 + 64)
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long *cnt_row(unsigned long long *base)
{
	return base + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (CNT_SLOTS - 1)) * CNT_N;
}
#endif

// One interval / list entry in HBM: x[0], x[1], x[2], info  (bwtintv_t, bwt.h:62)
struct __attribute__((aligned(32))) DevIntv { uint64_t x0, x1, x2, info; };

// Seed as produced by the SA-lookup kernel (mem_seed_t, bwa.h:121, plus the contig id of bwamem.c:293).
struct __attribute__((aligned(16))) DevSeed { int64_t rbeg; int32_t qbeg, len; int32_t score, rid; };

// Chain header as produced by the chaining kernel (mem_chain_t, bwa.h:135).
struct DevChain {
	int64_t pos;
	int32_t seed_off, n;     // seeds live at chain_seeds[read_seed_base + seed_off .. +n)
	int32_t rid, w, kept, first, is_alt;
	float frac_rep;
};

// Alignment region in HBM (mem_alnreg_t, bwa.h:145; same field order, no bit-fields)
struct DevReg {
	int64_t rb, re;
	float frac_rep;
	int32_t qb, qe, rid, score, truesc, sub, csub, sub_n, w, seedcov, seedlen0, n_comp, is_alt;
	int32_t pad;
};

#include <thread>
#include <vector>
// host helper: f(begin, end) on nt threads over contiguous chunks of [0, n)
template <class F> static inline void par_for_chunks(int64_t n, int nt, F f)
{
	if (nt <= 1 || n < 4096) { f((int64_t)0, n); return; }
	std::vector<std::thread> th;
	const int64_t per = (n + nt - 1) / nt;
	for (int t = 0; t < nt; ++t) { const int64_t b = t * per, e = b + per < n ? b + per : n; if (b < e) th.emplace_back(f, b, e); }
	for (auto &x : th) x.join();
}

struct HostIndex {           // host copy of a loaded index (index_io.cpp)
	bwahip_bwt_t bwt;
	bwahip_bns_t bns;
	uint8_t *pac;
	bool owned;
};
int bwahip_load_index_files(const char *prefix, HostIndex *out);   // index_io.cpp
void bwahip_free_host_index(HostIndex *h);
int bwahip_copy_host_index(const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac, HostIndex *out);

DevOpt make_dev_opt(const bwahip_opt_t *o);

// ---- launch wrappers (one per kernel translation unit) ----
struct SmemLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const uint8_t *seq; const int64_t *off;
	uint64_t *seq4; int seq4_stride;            // reads as 4-bit codes, seq4_stride 64-bit words per read (k_pack4)
	DevIntv *out; int *out_n; int cap;          // per read: out[read*cap .. ), out_n[read]
	DevIntv *raw; int *raw_n;                   // the same, unsorted: filled by k_smem / k_smem_heavy / k_smem3, sorted into `out` by k_intv_sort
	int *seed_cnt;                              // per read: number of SA look-ups chaining will do (bwamem.c:285-286)
	float *frac_rep_lrep;                       // unused slot (kept for layout stability)
	int *l_rep;                                 // per read: number of bwt_extend calls (diagnostic)
	DevIntv *scratch; int lcap;                 // per group: lcap 16-byte list spill entries
	unsigned int *queue;                        // work-queue head
	unsigned long long *counters;
	int *err;
	int groups_total;
	int *worst_n;                               // largest interval count of a read whose list overflowed `cap`
	int *heavy_list; unsigned int *heavy_n; int heavy_mult;   // reads handed to k_smem_heavy after heavy_mult*len extends (0: never)
};
int launch_smem(const SmemLaunch &a, int group_lanes, hipStream_t st);
int launch_pack4(const SmemLaunch &a, hipStream_t st);
int launch_smem_heavy(const SmemLaunch &a, hipStream_t st);
int launch_smem3(const SmemLaunch &a, hipStream_t st);
int launch_kmer_table(const DevIndex &ix, uint4 *tab, int K, hipStream_t st);   // fills kmer_off(K + 1) entries
int launch_kmer_check(const DevIndex &ix, int L, unsigned long long *bad, hipStream_t st);
int launch_intv_sort(const SmemLaunch &a, hipStream_t st);
int smem_default_groups(int group_lanes);

struct SeedLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const int64_t *off;
	const DevIntv *intv; const int *intv_n; int cap;
	const int64_t *seed_base;                   // exclusive scan of seed_cnt, n_reads+1 entries
	DevSeed *seeds;
	unsigned long long *counters;
};
int launch_seeds(const SeedLaunch &a, int64_t total_seeds, hipStream_t st);
int launch_bwt_planes(uint32_t *bwt, uint64_t n_words, hipStream_t st);   // the bases of every Occ block from 2-bit codes to bit planes, in place
int launch_sa_densify(const DevIndex &ix, uint64_t *dense, int to_intv, hipStream_t st);   // SA rows between the sampled ones, computed on the GPU

struct BtNodeOpaque { int w[48]; };              // sizeof(BtNode) in k_chain.hip (2 + 11 + 12 ints, pad, 11 x int64)
struct ChainWOpaque { int64_t pos; int a, b, c, d; int64_t e; int f, g, h, i; };
struct ChainLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const int64_t *off;
	const DevIntv *intv; const int *intv_n; int cap;
	const int64_t *seed_base; const DevSeed *seeds;
	// per-seed-slot scratch (indexed from seed_base[r])
	struct ChainWOpaque *cw_; int *nxt, *ord, *wts, *kept, *first, *keep_list;
	struct BtNodeOpaque *nodes_; int *stack;     // nodes: (seed_base>>2) + 4r ; stack: 256 ints per read
	// outputs
	DevChain *chains; DevSeed *chain_seeds; int *chain_n, *kept_seeds;
	// optional stage dump of the unfiltered chains (nullptr = off)
	DevChain *dbg_chains; DevSeed *dbg_seeds; int *dbg_chain_n;
	unsigned long long *counters;
	int *flt;                                    // 8 ints per seed slot: per-position data for k_chain_flt
	int *heavy_list, *heavy_count;               // reads whose overlap filter is deferred to k_chain_flt
	const int *seed_cnt; int *perm, *perm_counts;   // k_chain's launch order: reads with many seeds first, like sizes together (nullptr = identity)
	int *big_list, *big_count; int big_min, big_max, mid_max, glb_grid;   // reads with big_min < seeds <= big_max go to k_chain_big (nullptr: off)
};
int launch_chain(const ChainLaunch &a, hipStream_t st, hipStream_t st2, hipStream_t st3, hipEvent_t fork, hipEvent_t join, hipEvent_t join3);
int launch_chain_flt(const ChainLaunch &a, hipStream_t st);

struct ExtLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const uint8_t *seq; const int64_t *off;
	const int64_t *seed_base; const DevChain *chains; const DevSeed *chain_seeds; const int *chain_n;
	const int64_t *reg_base;                     // exclusive scan of kept_seeds
	DevReg *regs; int *reg_n;                    // final regions of read r at regs[reg_base[r] .. +reg_n[r])
	DevReg *dbg_regs; int *dbg_reg_n;            // optional: regions before mem_sort_dedup_patch
	DevReg *tmp_regs;                            // spare list of the same size as regs (sort gather)
	const int *kept_seeds; int *perm, *perm_counts;   // launch order: reads with many seeds first (nullptr = identity)
	int *srt;                                    // per-seed-slot scratch (sorted seed order), 2 ints per slot
	unsigned long long *counters; int *err;
	// k_extend_spec: best seed of chain ci of read r extended ahead of time into spec_regs[seed_base[r] + ci], for reads with
	// >= spec_min_chains chains (spec_regs == nullptr: off)
	DevReg *spec_regs; int2 *spec_items; int *spec_n; int spec_min_chains;
	int rank_sort_min;                           // dedup: lists at least this long try the wavefront rank sort first (shorter: one-lane introsort hides behind other wavefronts)
	int *redo_list, *redo_n;                     // reads k_extend hands to k_extend_big (reference window beyond the LDS window)
	int *dedup_list, *dedup_n;                   // reads k_extend leaves with more than one region: k_dedup sorts / dedups / patches them (three lists of n_reads: [0] the bulk, [1] the heavy reads, [2] what k_dedup_fast passes on to k_dedup)
	int subset;                                  // k_extend / k_dedup: 0 all reads, 1 the heavy ones (first two classes of the launch order), 2 the others; k_dedup 3: list 2
	uint8_t *big_t;                              // BWAHIP_EXT_BIG_GRID slabs of BWAHIP_EXT_BIG_T + 64 bytes
	int lds_window;                              // largest reference window k_extend keeps in LDS (<= its compiled MAXT)
};
constexpr int BWAHIP_EXT_BIG_GRID = 128, BWAHIP_EXT_BIG_T = 1 << 16;
int launch_extend(const ExtLaunch &a, int max_len, hipStream_t st, hipStream_t st2, hipEvent_t fork, hipEvent_t join);
int launch_order(int n, const int *keys, int t0, int t1, int t2, int *perm, int *counts, hipStream_t st);   // launch order by classes of keys[r], heaviest first
int launch_extend_spec(const ExtLaunch &a, int max_len, hipStream_t st);

// ---- finalisation on the GPU (k_final.hip): mark primary -> alignments (CIGAR, NM, MD, mapQ) -> SAM text ----
// Region with the fields finalisation adds (mem_alnreg_t, bwa.h:145-163, no bit-fields)
struct FinReg {
	int64_t rb, re; uint64_t hash; float frac_rep;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	int32_t pad;
};
// mem_aln_t (bwa.h:173-184); its CIGAR words and MD text live in the batch's text pool
struct DevAln {
	int64_t pos; int32_t rid, flag; uint32_t is_rev, is_alt, mapq, NM;
	int32_t n_cigar, score, sub, alt_sc;
	int64_t cigar_off;                           // byte offset of n_cigar uint32 words in the pool
	int64_t md_off; int32_t md_len, pad;
	int64_t pad2;                                // 80 bytes: elements of an array stay 16-byte aligned
};
static_assert(sizeof(DevAln) == 80 && sizeof(DevReg) == 80, "record sizes");
// what a region is needed for (FinLaunch::need)
enum { NEED_REC = 1 /* prints a SAM record */, NEED_XA = 2 /* listed in another record's XA tag */, NEED_H = 4 /* only as the mate information of the other end (h[i], bwamem_pair.c:404) */ };
struct DevPes { int low, high, failed, pad; double avg, std; };   // mem_pestat_t, bwa.h:167-171
struct PeRead;

struct FinLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const uint8_t *seq; const int64_t *off;
	int64_t n_processed;
	const double *logtab;
	// regions of mem_align1_core (k_extend): regs[reg_base[r] .. +reg_n[r])
	const DevReg *regs; const int64_t *reg_base; const int *reg_n;
	// mark primary: fregs at the same slots, in the reference's final order; n_pri per read; scratch
	FinReg *fregs, *fregs2; int *freg_n, *n_pri; int *scr;     // scr: 4 ints per region slot
	// paired end: the pairs mate rescue works on are finalised in a second launch, after their rescue (which runs beside the first launch).
	// subset 0: all reads; 1: all but the pairs flagged in resc_flag; 2: the pairs of resc_list only (grid = 2 x their number)
	const uint8_t *resc_flag; const int *resc_pairs; int subset;
	int read_lo;                                                // k_sam: first read of this launch (the write pass runs in two halves so that the download of the first overlaps the second)
	uint8_t *need; int *xa_owner;                // per region slot
	int *task_n, *rec_n;                         // per read: regions needing reg2aln; SAM records
	// alignment tasks
	const int64_t *task_base;                    // exclusive scan of task_n
	int2 *tasks;                                 // (read, region index)
	int *fast_list, *dp_list, *list_n;           // task ids without / with a DP (k_cigar<true> / <false>); list_n[2]: their lengths
	int *aln_of_reg;                             // per region slot: task id or -1
	DevAln *alns;
	uint8_t *pool; unsigned long long *pool_head; unsigned long long pool_cap;   // CIGAR / MD text: bump allocation
	int *redo_list, *redo_n; uint8_t *big_z;     // tasks whose backtrack matrix / window exceed LDS (k_cigar_big)
	unsigned *zslab;                             // k_cigar (DP tasks): one slab of backtrack cells per workgroup (cigar_zslab_bytes)
	// SAM text
	const uint8_t *qual; const int64_t *qual_off;   // qualities: read r at qual + qual_off[r], or qual_off[r] < 0: none ('*')
	const uint8_t *names; const int64_t *name_off; const uint8_t *comments; const int64_t *comment_off;   // per read (comments may be null)
	const uint8_t *ctg_names; const int *ctg_name_off; const uint8_t *ctg_anno; const int *ctg_anno_off;
	const uint8_t *rg_id; int rg_len;
	int *sam_len; const int64_t *sam_off; uint8_t *sam;        // per read length (pass 1), exclusive scan, text (pass 2)
	const DevAln **rec_list, **xa_list;          // per region slot: the read's record list / the XA members of the record being printed
	const PeRead *pe_read; DevPes pes[4];        // paired-end batches: per-read decisions of k_pair, insert-size statistics
	int *err;
};
int launch_mark_primary(const FinLaunch &a, bool plan, int n_listed, hipStream_t st);   // subset 2: n_listed pairs of resc_pairs
int launch_task_fill(const FinLaunch &a, hipStream_t st);
int launch_cigar(const FinLaunch &a, int n_fast, int n_dp, int max_len, hipStream_t st, hipStream_t st2, hipEvent_t fork, hipEvent_t join);
int launch_cigar_big(const FinLaunch &a, int grid, hipStream_t st);   // the tasks k_cigar listed in redo_list; big_z: grid slabs of cigar_big_slab_bytes()
size_t cigar_big_slab_bytes();
size_t cigar_zslab_bytes(int max_len, int n_dp);
int launch_sam(const FinLaunch &a, bool write, hipStream_t st, int read_lo = 0, int read_hi = -1);   // reads [read_lo, read_hi) (default: all)

// ---- paired-end stages on the GPU (k_pair.hip): insert-size histogram, mate rescue, pairing ----
// what the SAM stage needs to know about one read of a pair (mem_sam_pe, bwamem_pair.c:276-419)
struct PeRead {
	int mode;          // 1: the pair was paired (bwamem_pair.c:311-384); 0: each end printed like a single-end read with its mate attached (397-418)
	int h_reg;         // region whose alignment is h[i] (the record of the paired mode / the mate information), -1: unaligned
	int alt_reg;       // paired mode: ALT hit printed as supplementary (bwamem_pair.c:371-377), else -1
	int mapq;          // paired mode: q_se[i]
	int extra_flag;    // 0x1 | 0x2 (proper pair)
	int pad[3];
};
struct SwRes { int state /* 0 none, 1 queued, 2 done */, score, te, qe, score2, te2, tb, qb; };
struct PairLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const uint8_t *seq; const int64_t *off;
	int64_t n_processed;
	const double *logtab;
	const DevReg *regs; const int64_t *reg_base; const int *reg_n;   // mem_align1_core's regions (k_extend)
	unsigned *hist;                              // 4 x (max_ins + 1) insert-size counts (mem_pestat's isize[], as a histogram)
	DevPes pes[4];
	const double *pair_tab; int tab_off[4];      // per direction: .721*log(2*erfc(|ns|/sqrt2)) for dist = low..high, made by the host's libm
	// mate rescue
	int *nb;                                     // per read: regions within pen_unpaired of the best, capped at max_matesw (bwamem_pair.c:291-297)
	int *pe_cap; const int64_t *pe_base;         // per read: capacity of its list after rescue, and the scan of it
	DevReg *pe_regs; int *pe_n;                  // the lists mem_sam_pe works on
	DevReg *pe_tmp; void *pe_keys; int *pe_idx;  // sort scratch of a list at its slots: spare list, 16-byte keys, 2 ints per slot
	int *resc_list; int *resc_n;                 // pairs that need at least one Smith-Waterman
	// the alignments of the anchors that need one on the lists as mem_align1_core left them, done ahead of the sequential pass,
	// four per wavefront (k_matesw_sw): slot = sw_base[anchor's read] + 4 * anchor + orientation
	const int64_t *sw_base; int *sw_cnt; SwRes *sw_res; int *sw_tasks; int2 *sw_info; int *sw_n;
	int *sw_n8; int sw_cap;                      // the word kernel's tasks (mates of 250..256 bases x a): taken from the END of sw_tasks (sw_cap entries)
	uint8_t *slab; size_t slab_stride;           // k_matesw: per-workgroup global scratch (reference window, column maxima, long-query working set)
	unsigned int *queue;                         // k_matesw's work-queue heads (one per instantiation)
	unsigned long long *counters;                // [0] SW calls, [1] rescued regions
	// pairing
	const FinReg *fregs; const int *freg_n; const int *n_pri;   // after k_mark on the pe lists (fregs is written: sub / secondary updates of bwamem_pair.c:347-350, 359-365)
	FinReg *fregs_w;
	uint8_t *resc_flag; int subset;                           // see FinLaunch
	FinReg *fregs_tmp;                                         // k_mark's second region array: free by the time k_pair runs (sort space of heavy pairs)
	uint8_t *need; int *xa_owner; int *task_n, *rec_n; int *scr;
	PeRead *pe_read;
	int *err;
};
int launch_pestat(const PairLaunch &a, hipStream_t st);
int launch_pe_prepare(const PairLaunch &a, hipStream_t st);      // nb, pe_cap
int launch_pe_copy(const PairLaunch &a, hipStream_t st);         // copy lists into pe_regs, list the pairs that need rescue
int launch_matesw(const PairLaunch &a, int grid, hipStream_t st);
int launch_resc_order(const PairLaunch &a, int n_resc, int *scratch, hipStream_t st);   // rescue list by list length, longest first (scratch: 3 n_resc + 8 ints)
int launch_matesw_sw(const PairLaunch &a, int n_tasks, int n_tasks8, int max_len, hipStream_t st);   // max_len: longest read of the batch
int launch_pair(const PairLaunch &a, int n_listed, hipStream_t st);   // subset 2: n_listed pairs of resc_list
size_t matesw_slab_bytes(int64_t window);
int launch_sam_pe(const FinLaunch &a, bool write, hipStream_t st, int read_lo = 0, int read_hi = -1);

// K3b: mem_flt_chained_seeds on the chains k_chain / k_chain_flt left (k_seedsw.hip)
struct SeedSwLaunch {
	DevIndex ix; DevOpt opt;
	int n_reads; const uint8_t *seq; const int64_t *off;
	const int64_t *seed_base; DevChain *chains; DevSeed *chain_seeds; const int *chain_n; const int *kept_seeds;
	const double *logtab;                        // log(i) for i < BWAHIP_LOGTAB_N, computed by the host's libm
};
int launch_seed_sw(const SeedSwLaunch &a, hipStream_t st);
int launch_kat_align(const DevOpt &opt, int n, int byte_mode, const int *items, const int *params, const uint8_t *q, const int64_t *qoff,
                     const uint8_t *t, const int64_t *toff, uint8_t *wsp, size_t wsp_stride, int *out7, hipStream_t st);
constexpr int BWAHIP_LOGTAB_N = 65536;

int launch_kat_ksw(const DevOpt &opt, int n, const int *params, const uint8_t *q, const int64_t *qoff, const uint8_t *t, const int64_t *toff,
                   int *out6, hipStream_t st);
int launch_kat_isort(int n, int mode, const void *keys16, int *idx_par, int *idx_seq, int *work, int *status, hipStream_t st);
size_t kat_isort_work_ints(int n);
int launch_kat_occ4(const DevIndex &ix, int n, const uint64_t *k, uint64_t *out, hipStream_t st);
int launch_kat_sa(const DevIndex &ix, int n, const uint64_t *k, uint64_t *out, hipStream_t st);
int launch_kat_extend(const DevIndex &ix, int n, const uint64_t *ik3, const int *is_back, uint64_t *ok12, hipStream_t st);
