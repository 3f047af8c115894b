// Context of libbwahip (one GPU, one index resident in HBM, the batch buffers) -- shared by the host translation units
// runtime.hip (hot path sequencing, C ABI) and final_rt.hip (finalisation / SAM sequencing).
#pragma once
#include "bwahip_internal.h"
#include <string>

// ------------------------------------------------------------------ small device helpers
struct DevBuf {
	void *p = nullptr; size_t cap = 0; bool ext = false;   // ext: caller-owned device memory (bwahip_batch_attach)
	void adopt(void *dev, size_t bytes) { release(); p = dev; cap = bytes; ext = true; }
	int ensure(size_t bytes)
	{
		if (bytes <= cap && !ext) return 0;
		if (p && !ext) (void)hipFree(p);
		p = nullptr; cap = 0; ext = false;
		size_t want = bytes + bytes / 8 + 256;
		if (hipMalloc(&p, want) != hipSuccess) { fprintf(stderr, "[bwahip] hipMalloc(%zu) failed\n", want); return BWAHIP_ENOMEM; }
		cap = want;
		return 0;
	}
	void release() { if (p && !ext) (void)hipFree(p); p = nullptr; cap = 0; ext = false; }
	template <class T> T *as() const { return (T*)p; }
};


// pinned host staging buffer (grows, never shrinks)
struct HostBuf {
	void *p = nullptr; size_t cap = 0;
	int ensure(size_t bytes)
	{
		if (bytes <= cap) return 0;
		if (p) (void)hipHostFree(p);
		p = nullptr; cap = 0;
		const size_t want = bytes + bytes / 8 + 4096;
		if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { fprintf(stderr, "[bwahip] hipHostMalloc(%zu) failed\n", want); p = nullptr; return BWAHIP_ENOMEM; }
		cap = want;
		return 0;
	}
	void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Tuning knobs (hand-off thresholds of the heavy-read kernels).  Read from the environment ONCE, when the context is
// created; bwahip_ctx_tune changes them afterwards (tests force every hand-off kernel onto ordinary reads that way).
struct Knobs {
	int intv_cap = 96;          // BWAHIP_INTV_CAP: initial per-read interval capacity (grown on overflow)
	int smem_lanes = 1;         // BWAHIP_SMEM_LANES: lanes per read in k_smem (1, 2, 4, 8)
	int sa_intv = 1;            // BWAHIP_SA_INTV: interval of the SA table in HBM (1: every row, 8 bytes each; the index files' own interval or more: the files' table as it is); a table that would take over a quarter of the free HBM is built at the next interval that fits
	int kmer_k = 14;            // BWAHIP_KMER_K: the interval table holds the bi-intervals of all strings of up to kmer_k bases (16 bytes each, 4^k of them per length: 14 -> 5.7 GB, 15 -> 23 GB, 16 -> 92 GB; 0 or 1: no table); never longer than log4 of the text, nor than a quarter of the free HBM
	int heavy_mult = -1;        // BWAHIP_HEAVY_MULT: hand a read to k_smem_heavy after heavy_mult x len extends (0: never; -1: 10 for reads up to 200 bases, 30 above -- a 250 bp read at 5 % error needs 2 500 extends on average, and the hand-off is for the outliers)
	int chain_mid_max = 1536, chain_big_max = 3200, chain_glb_grid = 2048;   // BWAHIP_CHAIN_MID_MAX / _BIG_MAX / _GLB_GRID: seeds up to which a read's B-tree lives in 77 KB / 150 KB of LDS (beyond: in global memory), and the workgroups of that last kernel
	int chain_big_min = 512;    // BWAHIP_CHAIN_BIG_MIN: seeds above which a wavefront-per-read chaining kernel takes the read (< 0: off)
	int rank_sort_min = 2;      // BWAHIP_RANK_SORT_MIN: dedup lists at least this long are sorted by the whole wavefront (shorter: the one-lane restatement of ks_introsort)
	int spec_min_chains = 16;   // BWAHIP_SPEC_MIN_CHAINS: chains from which k_extend_spec extends ahead of time (0: off)
	int ext_lds_window = 1 << 30;   // BWAHIP_EXT_LDS_WINDOW: reference windows above this go to k_extend_big (tests; default = the compiled LDS window)
	int gpu_final = 1;          // BWAHIP_GPU_FINAL: 0 = finalisation of single-end batches on host threads (host_final.cpp) instead of the GPU kernels
	int gpu_pair = 1;           // BWAHIP_GPU_PAIR: 0 = paired-end batches finalised on host threads (mate rescue, pairing, SAM)
	int verbose = 0;            // BWAHIP_VERBOSE
	int e2e_log = 0;            // BWAHIP_E2E_LOG: one line of phase timings per bwahip_process_seqs call
	const char *dump_ext = nullptr;   // BWAHIP_DUMP_EXT (diagnostic)
	void from_env()
	{
		auto geti = [](const char *k, int &v) { if (const char *e = getenv(k)) v = atoi(e); };
		geti("BWAHIP_INTV_CAP", intv_cap); geti("BWAHIP_SMEM_LANES", smem_lanes); geti("BWAHIP_HEAVY_MULT", heavy_mult); geti("BWAHIP_SA_INTV", sa_intv); geti("BWAHIP_KMER_K", kmer_k);
		geti("BWAHIP_CHAIN_BIG_MIN", chain_big_min); geti("BWAHIP_CHAIN_MID_MAX", chain_mid_max); geti("BWAHIP_CHAIN_BIG_MAX", chain_big_max); geti("BWAHIP_CHAIN_GLB_GRID", chain_glb_grid); geti("BWAHIP_RANK_SORT_MIN", rank_sort_min); geti("BWAHIP_SPEC_MIN_CHAINS", spec_min_chains); geti("BWAHIP_EXT_LDS_WINDOW", ext_lds_window); geti("BWAHIP_GPU_FINAL", gpu_final); geti("BWAHIP_GPU_PAIR", gpu_pair);
		verbose = getenv("BWAHIP_VERBOSE") != nullptr;
		e2e_log = getenv("BWAHIP_E2E_LOG") != nullptr;
		dump_ext = getenv("BWAHIP_DUMP_EXT");
		if (intv_cap < 2) intv_cap = 2;
	}
};

// offsets of one host batch on its way to HBM (bwahip_process_seqs)
struct BatchText {
	std::vector<int64_t> off, qoff, noff, coff;
	int64_t qtot = 0;
	bool any_comment = false;
	size_t sz_codes = 0, sz_qual = 0, sz_names = 0, sz_comm = 0;
};

struct bwahip_ctx {
	BatchText batch_text;
	bool want_host_sam_off = false;      // run_final copies the SAM offsets to h_sam_off ahead of the write pass (bwahip_process_seqs)
	std::vector<int64_t> h_sam_off;      // offsets of the reads' SAM text in h_sam (bwahip_process_seqs / _text)
	bool external_index = false;
	bool index_resident = false;         // d_bwt / d_sa / d_pac were filled before ctx_setup (bwahip_init_rccl)
	Knobs knobs;
	std::string rg_id;                   // read-group id appended as RG:Z: to every record (bwa_rg_id, bwa.c:44); empty = none
	DevBuf d_logtab;                     // log(i), i < BWAHIP_LOGTAB_N, from the host's libm (bwamem.c:607, 974-981)         // index arrays live in caller-owned HBM (bwahip_init_device)
	int device = 0;
	hipStream_t stream_copy = nullptr;   // uploads that run beside the kernels (bwahip_process_seqs: names / qualities during the hot path)
	hipEvent_t ev_sam_half = nullptr; int sam_half_reads = 0;   // run_final: recorded when the SAM text of reads [0, sam_half_reads) is written
	hipEvent_t ev_slice[8] = {};         // bwahip_process_seqs: one per slice of the SAM download
	hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;   // stream2/3: kernels that run beside the main one (k_chain_big)
	hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
	HostIndex host = {};                 // host copy: contig table + packed reference (always owned); FM-index arrays only when loaded from files
	DevIndex ix;
	DevBuf d_bwt, d_sa, d_pac, d_anns;
	DevBuf d_bwtp;                       // bit-plane copy of a caller-owned BWT (bwahip_init_device); otherwise d_bwt itself is re-laid in place
	bool bwt_is_planes = false;          // d_bwt already holds the bit-plane layout (bwahip_ctx_clone_on: copied from a context's array)
	DevBuf d_kmer;                       // the interval table of the BWT search (launch_kmer_table); clones read their source's
	const bwahip_ctx *share_from = nullptr;   // bwahip_ctx_clone: the context whose index arrays this one reads
	DevBuf d_sa_dense;                   // the SA table the kernels read when it is denser than the files' (launch_sa_densify); owned by the context that built it
	// batch state
	int n_reads = 0, max_len = 0;
	int64_t total_bases = 0;
	DevBuf d_seq, d_off, d_seq4, d_smem_heavy, d_raw, d_raw_n;
	DevBuf d_intv, d_intv_n, d_seed_cnt, d_lrep, d_seed_base, d_seeds, d_scratch;
	DevBuf d_misc;                       // CNT_SLOTS rows of CNT_N counters (u64), then queue (4 x u32), err (i32)
	// K3/K4 working set (sized from the seed count of the batch)
	DevBuf d_cw, d_nxt, d_ord, d_wts, d_kept, d_first, d_keep, d_nodes, d_stack;
	DevBuf d_chains, d_chain_seeds, d_chain_n, d_kept_seeds, d_reg_base, d_regs, d_tmp_regs, d_reg_n, d_srt;
	DevBuf d_dbg_chains, d_dbg_seeds, d_dbg_chain_n, d_dbg_regs, d_dbg_reg_n, d_flt, d_heavy, d_perm, d_spec_regs, d_spec_items, d_scan, d_chain_big, d_redo, d_big_t, d_dedup, d_cperm;
	// finalisation on the GPU (final_rt.hip)
	DevBuf d_ctg_names, d_ctg_name_off, d_ctg_anno, d_ctg_anno_off, d_rg;      // contig names / annotations (SAM RNAME, XR), read-group id
	DevBuf d_qual, d_qual_off, d_names, d_name_off, d_comments, d_comment_off; // per-batch text inputs of the SAM kernels
	DevBuf d_fregs, d_fregs2, d_fscr, d_need, d_xa_owner, d_freg_n, d_npri, d_task_n, d_rec_n, d_task_base, d_tasks, d_aln_of_reg, d_alns;
	DevBuf d_hist, d_pair_tab, d_nb, d_pe_cap, d_pe_base, d_pe_regs, d_pe_n, d_pe_tmp, d_pe_keys, d_pe_idx, d_resc, d_ms_slab, d_pe_read, d_sw_cnt, d_sw_base, d_sw_res, d_sw_tasks, d_sw_info;   // paired-end stages
	bwahip_pestat_t last_pes[4];         // insert-size statistics of the last paired-end batch
	unsigned long long last_sw_tasks = 0;   // alignments k_matesw_sw ran ahead of the list logic (BWAHIP_PE_LOG)
	unsigned long long last_pe_counters[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // mate-rescue alignments run / regions added / most per pair / pairs rescued
	DevBuf d_task_lists;                 // k_cigar's two work lists (no-DP tasks, DP tasks)
	DevBuf d_resc_flag;                  // one byte per pair: mate rescue works on it (finalised by the second k_mark / k_pair launch)
	DevBuf d_zslab;                      // k_cigar's backtrack slabs
	DevBuf d_resc_ord;                   // scratch of the rescue list's ordering
	DevBuf d_pool, d_fmisc, d_fredo, d_bigz, d_rec_list, d_xa_list, d_sam_len, d_sam_off, d_sam;
	HostBuf h_stage, h_sam, h_sam2;       // pinned staging: batch text in, SAM text out (two buffers taken in turn by bwahip_process_seqs_text)
	int sam_flip = 0;
	int64_t total_tasks = 0, total_sam = 0;
	size_t pool_cap = 0;
	float final_ms[8] = { 0 };           // k_mark, k_cigar, k_sam(size), k_sam(write) of the last run
	int intv_cap = 96;                   // current capacity (starts at knobs.intv_cap, grows on overflow)
	int64_t total_seeds = 0, total_regs = 0;
	hipEvent_t ev[24];
	float last_ms[24];
};


extern "C" int ctx_setup(bwahip_ctx *c, const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac);   // streams, tables, DevIndex
int launch_scan(const int *in, int64_t *out, int n, DevBuf &tmp, hipStream_t st);   // exclusive scan int32 -> int64, n+1 outputs
int launch_nt4(uint8_t *seq, int64_t n, hipStream_t st);   // runtime.hip: ASCII / codes -> codes 0..4 in place (nst_nt4_table)
int dev_upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st);
int run_pipeline(bwahip_ctx *c, const bwahip_opt_t *opt, bool timed, bool dump);      // the hot path over the uploaded batch
int run_final(bwahip_ctx *c, const bwahip_opt_t *opt, int64_t n_processed, const bwahip_pestat_t *pes0, bool timed);   // regions in HBM -> SAM text in HBM (SE, or PE when opt->flag has MEM_F_PE)
int final_setup(bwahip_ctx *c);                                                       // contig name tables for the SAM kernels
