// Host sequencing of the GPU finalisation (regions of mem_align1_core in HBM -> SAM text in HBM) and the C ABI entry
// bwahip_process_seqs == mem_process_seqs (bwamem.h:69).  Kernels: k_final.hip (mark primary, selection, CIGAR/NM/MD/mapQ),
// k_sam.hip (SAM text), k_pair.hip (insert sizes, mate rescue, pairing).
#include "ctx_internal.h"
#include <atomic>
#include <thread>
#include <math.h>
#include <algorithm>
#include <chrono>

#include <dlfcn.h>
// The host finalisation is test infrastructure (csrc/host_final.cpp, built as libbwahip_hostfinal.so beside this library) and not linked
// in: the knobs gpu_final = 0 / gpu_pair = 0 load it on demand.  Without it those knobs are an error -- there is no CPU path in the product.
typedef int (*host_final_fn)(bwahip_ctx*, const bwahip_opt_t*, int64_t, int, bwahip_seq_t*, const bwahip_pestat_t*);
static host_final_fn load_host_final()
{
	static host_final_fn fn = nullptr;
	static bool tried = false;
	if (tried) return fn;
	tried = true;
	Dl_info di;
	std::string path = "libbwahip_hostfinal.so";
	if (dladdr((const void*)&bwahip_version, &di) && di.dli_fname) {
		const std::string self = di.dli_fname;
		const size_t sl = self.rfind('/');
		if (sl != std::string::npos) path = self.substr(0, sl + 1) + path;
	}
	void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
	if (h) fn = (host_final_fn)dlsym(h, "bwahip_process_seqs_host");
	if (!fn) fprintf(stderr, "[bwahip] gpu_final / gpu_pair = 0 need the test library %s (%s)\n", path.c_str(), h ? "symbol missing" : dlerror());
	return fn;
}

// contig names and annotations as flat byte tables (RNAME, SA / XA entries, XR)
int final_setup(bwahip_ctx *c)
{
	const bwahip_bns_t &bns = c->host.bns;
	std::vector<uint8_t> names, anno;
	std::vector<int> noff(bns.n_seqs + 1, 0), aoff(bns.n_seqs + 1, 0);
	for (int i = 0; i < bns.n_seqs; ++i) {
		const char *nm = bns.anns[i].name ? bns.anns[i].name : "", *an = bns.anns[i].anno ? bns.anns[i].anno : "";
		names.insert(names.end(), nm, nm + strlen(nm)); noff[i + 1] = (int)names.size();
		anno.insert(anno.end(), an, an + strlen(an)); aoff[i + 1] = (int)anno.size();
	}
	names.resize(names.size() + 64, 0); anno.resize(anno.size() + 64, 0);
	int rc;
	if ((rc = dev_upload(c->d_ctg_names, names.data(), names.size(), c->stream)) || (rc = dev_upload(c->d_ctg_name_off, noff.data(), noff.size() * 4, c->stream)) ||
	    (rc = dev_upload(c->d_ctg_anno, anno.data(), anno.size(), c->stream)) || (rc = dev_upload(c->d_ctg_anno_off, aoff.data(), aoff.size() * 4, c->stream)) ||
	    (rc = c->d_fmisc.ensure(256))) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return 0;
}

// mem_pestat (bwamem_pair.c:72-134) from the insert-size histogram the GPU collected: the sorted list of the reference is the
// histogram read in ascending order, so percentiles, mean and (added in the same order) the sum of squares come out the same.
static void pestat_from_hist(const bwahip_opt_t *opt, const std::vector<unsigned> &hist, bwahip_pestat_t pes[4])
{
	const int W = opt->max_ins + 1;
	uint64_t cnt[4] = { 0, 0, 0, 0 };
	memset(pes, 0, 4 * sizeof(bwahip_pestat_t));
	for (int d = 0; d < 4; ++d) for (int v = 1; v < W; ++v) cnt[d] += hist[(size_t)d * W + v];
	for (int d = 0; d < 4; ++d) {
		bwahip_pestat_t *r = &pes[d];
		const unsigned *h = &hist[(size_t)d * W];
		const uint64_t nq = cnt[d];
		if (nq < 10) { r->failed = 1; continue; }                    // MIN_DIR_CNT
		auto at = [&](uint64_t idx) { uint64_t run = 0; for (int v = 1; v < W; ++v) { run += h[v]; if (run > idx) return v; } return W - 1; };
		const int p25 = at((uint64_t)(int)(.25 * nq + .499)), p75 = at((uint64_t)(int)(.75 * nq + .499));
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);              // OUTLIER_BOUND
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		uint64_t x = 0;
		r->avg = 0;
		for (int v = 1; v < W; ++v) if (v >= r->low && v <= r->high) { r->avg += (double)v * h[v]; x += h[v]; }   // integers: exact in any order
		r->avg /= x;
		r->std = 0;
		for (int v = 1; v < W; ++v) if (v >= r->low && v <= r->high) { const double t = ((double)v - r->avg) * ((double)v - r->avg); for (unsigned k = 0; k < h[v]; ++k) r->std += t; }
		r->std = sqrt(r->std / x);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);              // MAPPING_BOUND
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);      // MAX_STDDEV
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
	}
	uint64_t mx = 0;
	for (int d = 0; d < 4; ++d) mx = mx > cnt[d] ? mx : cnt[d];
	for (int d = 0; d < 4; ++d) if (pes[d].failed == 0 && cnt[d] < (int)mx * 0.05) pes[d].failed = 1;   // MIN_DIR_RATIO
}

// The paired-end stages before mark-primary: insert-size statistics, mate rescue.  Leaves the per-read lists in d_pe_regs.
// Insert sizes, the lists of both ends, and mate rescue.  The rescue kernels (a handful of pairs, long dependent chains: the GPU is nearly idle
// under them) go to the second stream and are NOT waited for: run_final finalises all other pairs meanwhile and the rescued ones after ev_join.
static int run_pe_rescue(bwahip_ctx *c, const bwahip_opt_t *opt, const DevOpt &dopt, int64_t n_processed, const bwahip_pestat_t *pes0, PairLaunch &pl, int &n_resc_out)
{
	const int n = c->n_reads;
	int rc;
	memset(&pl, 0, sizeof pl);
	pl.ix = c->ix; pl.opt = dopt; pl.n_reads = n; pl.seq = c->d_seq.as<uint8_t>(); pl.off = c->d_off.as<int64_t>(); pl.n_processed = n_processed;
	pl.logtab = c->d_logtab.as<double>();
	pl.regs = c->d_regs.as<DevReg>(); pl.reg_base = c->d_reg_base.as<int64_t>(); pl.reg_n = c->d_reg_n.as<int>();
	unsigned long long *fm = c->d_fmisc.as<unsigned long long>();
	pl.err = (int*)(fm + 2); pl.resc_n = (int*)(fm + 4); pl.counters = fm + 5; pl.sw_n = (int*)(fm + 14); pl.queue = (unsigned int*)(fm + 16); pl.sw_n8 = (int*)(fm + 18);
	bwahip_pestat_t pes[4];
	if (pes0) memcpy(pes, pes0, sizeof pes);
	else {
		if (opt->max_ins < 1 || opt->max_ins > (1 << 24)) return BWAHIP_EINVAL;
		const size_t hb = (size_t)4 * (opt->max_ins + 1) * 4;
		if ((rc = c->d_hist.ensure(hb))) return rc;
		HIP_TRY(hipMemsetAsync(c->d_hist.p, 0, hb, c->stream));
		pl.hist = c->d_hist.as<unsigned>();
		if ((rc = launch_pestat(pl, c->stream))) return rc;
		std::vector<unsigned> h((size_t)4 * (opt->max_ins + 1));
		HIP_TRY(hipMemcpyAsync(h.data(), c->d_hist.p, hb, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		pestat_from_hist(opt, h, pes);
	}
	memcpy(c->last_pes, pes, sizeof pes);
	// per direction: .721 * log(2 * erfc(|ns| / sqrt 2)) for every admissible distance (bwamem_pair.c:243-244), by the host's libm
	std::vector<double> tab;
	int64_t widest = 0;
	for (int d = 0; d < 4; ++d) {
		pl.pes[d].low = pes[d].low; pl.pes[d].high = pes[d].high; pl.pes[d].failed = pes[d].failed; pl.pes[d].pad = 0; pl.pes[d].avg = pes[d].avg; pl.pes[d].std = pes[d].std;
		pl.tab_off[d] = (int)tab.size();
		if (pes[d].failed || pes[d].high < pes[d].low) continue;
		if ((int64_t)pes[d].high - pes[d].low > (1 << 26)) return BWAHIP_EINVAL;
		widest = std::max<int64_t>(widest, (int64_t)pes[d].high - pes[d].low);
		for (int64_t dist = pes[d].low; dist <= pes[d].high; ++dist) {
			const double ns = (dist - pes[d].avg) / pes[d].std;
			tab.push_back(.721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)));
		}
	}
	tab.push_back(0.);
	if ((rc = dev_upload(c->d_pair_tab, tab.data(), tab.size() * 8, c->stream))) return rc;
	pl.pair_tab = c->d_pair_tab.as<double>();
	// list capacities after rescue
	if ((rc = c->d_nb.ensure((size_t)n * 4)) || (rc = c->d_pe_cap.ensure((size_t)n * 4)) || (rc = c->d_pe_base.ensure((size_t)(n + 1) * 8)) || (rc = c->d_pe_n.ensure((size_t)n * 4)) ||
	    (rc = c->d_resc.ensure((size_t)(n / 2 + 4) * 4)) || (rc = c->d_resc_flag.ensure((size_t)(n / 2 + 4))) || (rc = c->d_sw_cnt.ensure((size_t)n * 4)) || (rc = c->d_sw_base.ensure((size_t)(n + 1) * 8))) return rc;
	pl.sw_cnt = c->d_sw_cnt.as<int>(); pl.sw_base = c->d_sw_base.as<int64_t>();
	pl.nb = c->d_nb.as<int>(); pl.pe_cap = c->d_pe_cap.as<int>(); pl.pe_base = c->d_pe_base.as<int64_t>(); pl.pe_n = c->d_pe_n.as<int>(); pl.resc_list = c->d_resc.as<int>(); pl.resc_flag = c->d_resc_flag.as<uint8_t>();
	HIP_TRY(hipMemsetAsync(c->d_resc_flag.p, 0, (size_t)(n / 2 + 4), c->stream));
	if ((rc = launch_pe_prepare(pl, c->stream))) return rc;
	if ((rc = launch_scan(pl.pe_cap, c->d_pe_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	if ((rc = launch_scan(pl.sw_cnt, c->d_sw_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	int64_t cap = 0, n_slots = 0;
	HIP_TRY(hipMemcpyAsync(&cap, c->d_pe_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(&n_slots, c->d_sw_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	{
		const size_t S = (size_t)(n_slots ? n_slots : 1);
		if ((rc = c->d_sw_res.ensure(S * sizeof(SwRes))) || (rc = c->d_sw_tasks.ensure(S * 4)) || (rc = c->d_sw_info.ensure(S * 8))) return rc;
		HIP_TRY(hipMemsetAsync(c->d_sw_res.p, 0, S * sizeof(SwRes), c->stream));
		pl.sw_res = c->d_sw_res.as<SwRes>(); pl.sw_tasks = c->d_sw_tasks.as<int>(); pl.sw_info = c->d_sw_info.as<int2>(); pl.sw_cap = (int)S;
	}
	const size_t R = (size_t)(cap ? cap : 1);
	c->total_regs = cap;                                          // from here on the region slots are the paired-end ones
	if ((rc = c->d_pe_regs.ensure(R * sizeof(DevReg))) || (rc = c->d_pe_tmp.ensure(R * sizeof(DevReg))) || (rc = c->d_pe_keys.ensure(R * 16)) || (rc = c->d_pe_idx.ensure(R * 8))) return rc;
	pl.pe_regs = c->d_pe_regs.as<DevReg>(); pl.pe_tmp = c->d_pe_tmp.as<DevReg>(); pl.pe_keys = c->d_pe_keys.p; pl.pe_idx = c->d_pe_idx.as<int>();
	if ((rc = launch_pe_copy(pl, c->stream))) return rc;
	int n_resc = 0, n_sw_tasks = 0, n_sw_tasks8 = 0;
	HIP_TRY(hipMemcpyAsync(&n_sw_tasks8, pl.sw_n8, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(&n_resc, pl.resc_n, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(&n_sw_tasks, pl.sw_n, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	n_resc_out = n_resc;
	c->last_pe_counters[3] = (unsigned long long)n_resc;
	c->last_sw_tasks = (unsigned long long)n_sw_tasks + (unsigned long long)n_sw_tasks8;
	if (n_resc > 0 || n_sw_tasks > 0 || n_sw_tasks8 > 0) {
		HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
		HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
		if ((rc = launch_matesw_sw(pl, n_sw_tasks, n_sw_tasks8, c->max_len, c->stream2))) return rc;   // the alignments against the unrescued lists, all at once
		if (n_resc > 0) {
			if ((rc = c->d_resc_ord.ensure(((size_t)3 * n_resc + 8) * 4)) || (rc = launch_resc_order(pl, n_resc, c->d_resc_ord.as<int>(), c->stream2))) return rc;
			const int grid = std::min(n_resc, 2048);
			pl.slab_stride = (matesw_slab_bytes(widest + c->max_len) + 255) & ~(size_t)255;
			if ((rc = c->d_ms_slab.ensure(pl.slab_stride * (size_t)grid))) return rc;
			pl.slab = c->d_ms_slab.as<uint8_t>();
			HIP_TRY(hipMemsetAsync(pl.queue, 0, 16, c->stream2));
			if ((rc = launch_matesw(pl, grid, c->stream2))) return rc;
		}
		HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
	}
	return 0;
}

// K6 -> K9 over the batch run_pipeline left in HBM.  The text inputs (d_qual, d_names, ...) must be uploaded.
int run_final(bwahip_ctx *c, const bwahip_opt_t *opt, int64_t n_processed, const bwahip_pestat_t *pes0, bool timed)
{
	const int n = c->n_reads;
	const bool pe = (opt->flag & BWAHIP_F_PE) != 0;
	c->total_sam = 0; c->total_tasks = 0;
	if (n == 0) return 0;
	if (pe && (n & 1)) return BWAHIP_EINVAL;
	int rc;
	HIP_TRY(hipMemsetAsync(c->d_fmisc.p, 0, 256, c->stream));
	if (timed) HIP_TRY(hipEventRecord(c->ev[20], c->stream));
	PairLaunch pl;
	const DevOpt dopt_pe = make_dev_opt(opt);
	int n_resc = 0;
	if (pe && (rc = run_pe_rescue(c, opt, dopt_pe, n_processed, pes0, pl, n_resc))) return rc;
	const size_t R = (size_t)(c->total_regs ? c->total_regs : 1);
	if ((rc = c->d_fregs.ensure(R * sizeof(FinReg))) || (rc = c->d_fregs2.ensure(R * sizeof(FinReg))) || (rc = c->d_fscr.ensure(R * 16)) || (rc = c->d_need.ensure(R)) ||
	    (rc = c->d_xa_owner.ensure(R * 4)) || (rc = c->d_aln_of_reg.ensure(R * 4)) || (rc = c->d_rec_list.ensure(R * 8)) || (rc = c->d_xa_list.ensure(R * 8)) ||
	    (rc = c->d_freg_n.ensure((size_t)n * 4)) || (rc = c->d_npri.ensure((size_t)n * 4)) || (rc = c->d_task_n.ensure((size_t)n * 4)) || (rc = c->d_rec_n.ensure((size_t)n * 4)) ||
	    (rc = c->d_task_base.ensure((size_t)(n + 1) * 8)) || (rc = c->d_sam_len.ensure((size_t)n * 4)) || (rc = c->d_sam_off.ensure((size_t)(n + 1) * 8))) return rc;
	// rg id
	{
		std::string rg = c->rg_id;
		rg.resize(rg.size() + 64, 0);
		if ((rc = dev_upload(c->d_rg, rg.data(), rg.size(), c->stream))) return rc;
	}
	FinLaunch f;
	memset(&f, 0, sizeof f);
	f.ix = c->ix; f.opt = make_dev_opt(opt); f.n_reads = n; f.seq = c->d_seq.as<uint8_t>(); f.off = c->d_off.as<int64_t>();
	f.n_processed = n_processed; f.logtab = c->d_logtab.as<double>();
	f.regs = c->d_regs.as<DevReg>(); f.reg_base = c->d_reg_base.as<int64_t>(); f.reg_n = c->d_reg_n.as<int>();
	if (pe) { f.regs = c->d_pe_regs.as<DevReg>(); f.reg_base = c->d_pe_base.as<int64_t>(); f.reg_n = c->d_pe_n.as<int>(); }
	f.fregs = c->d_fregs.as<FinReg>(); f.fregs2 = c->d_fregs2.as<FinReg>(); f.freg_n = c->d_freg_n.as<int>(); f.n_pri = c->d_npri.as<int>(); f.scr = c->d_fscr.as<int>();
	f.need = c->d_need.as<uint8_t>(); f.xa_owner = c->d_xa_owner.as<int>(); f.task_n = c->d_task_n.as<int>(); f.rec_n = c->d_rec_n.as<int>();
	f.task_base = c->d_task_base.as<int64_t>(); f.aln_of_reg = c->d_aln_of_reg.as<int>();
	f.rec_list = c->d_rec_list.as<const DevAln*>(); f.xa_list = c->d_xa_list.as<const DevAln*>();
	unsigned long long *pool_head = c->d_fmisc.as<unsigned long long>();
	f.pool_head = pool_head; f.redo_n = (int*)(pool_head + 1); f.err = (int*)(pool_head + 2);
	f.qual = c->d_qual.as<uint8_t>(); f.qual_off = c->d_qual_off.as<int64_t>(); f.names = c->d_names.as<uint8_t>(); f.name_off = c->d_name_off.as<int64_t>();
	f.comments = c->d_comments.p ? c->d_comments.as<uint8_t>() : nullptr; f.comment_off = c->d_comment_off.as<int64_t>();
	f.ctg_names = c->d_ctg_names.as<uint8_t>(); f.ctg_name_off = c->d_ctg_name_off.as<int>(); f.ctg_anno = c->d_ctg_anno.as<uint8_t>(); f.ctg_anno_off = c->d_ctg_anno_off.as<int>();
	f.rg_id = c->d_rg.as<uint8_t>(); f.rg_len = (int)c->rg_id.size();
	f.sam_len = c->d_sam_len.as<int>(); f.sam_off = c->d_sam_off.as<int64_t>();
	if (timed) HIP_TRY(hipEventRecord(c->ev[15], c->stream));
	// paired end: first every pair mate rescue does not touch, while the rescue kernels run on the second stream; then, once those are done, their pairs
	if (pe) { f.resc_flag = pl.resc_flag; f.resc_pairs = pl.resc_list; f.subset = 1; }
	if ((rc = launch_mark_primary(f, !pe, 0, c->stream))) return rc;
	if (pe) {                                                     // mem_pair + the decisions of mem_sam_pe
		if ((rc = c->d_pe_read.ensure((size_t)n * sizeof(PeRead)))) return rc;
		pl.fregs = f.fregs; pl.fregs_w = f.fregs; pl.fregs_tmp = f.fregs2; pl.freg_n = f.freg_n; pl.n_pri = f.n_pri; pl.need = f.need; pl.xa_owner = f.xa_owner;
		pl.task_n = f.task_n; pl.rec_n = f.rec_n; pl.scr = f.scr; pl.pe_read = c->d_pe_read.as<PeRead>();
		pl.subset = 1;
		if ((rc = launch_pair(pl, 0, c->stream))) return rc;
		HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));      // the rescue kernels (an event not recorded in this batch is complete: no wait)
		if (n_resc > 0) {
			f.subset = 2; pl.subset = 2;
			if ((rc = launch_mark_primary(f, false, n_resc, c->stream))) return rc;
			if ((rc = launch_pair(pl, n_resc, c->stream))) return rc;
		}
		f.subset = 0; pl.subset = 0;
		HIP_TRY(hipMemcpyAsync(c->last_pe_counters, pl.counters, 72, hipMemcpyDeviceToHost, c->stream));   // [0..3] as documented; [4..8]: ticks of 10 ns in the window fetch, the alignments and the list clean-up of k_matesw, its longest pair, alignments it ran itself
		f.pe_read = pl.pe_read;
		memcpy(f.pes, pl.pes, sizeof f.pes);
	}
	if ((rc = launch_scan(c->d_task_n.as<int>(), c->d_task_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	if (timed) HIP_TRY(hipEventRecord(c->ev[16], c->stream));
	int64_t T = 0;
	HIP_TRY(hipMemcpyAsync(&T, c->d_task_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->total_tasks = T;
	const size_t Tn = (size_t)(T ? T : 1);
	if ((rc = c->d_tasks.ensure(Tn * 8)) || (rc = c->d_alns.ensure(Tn * sizeof(DevAln))) || (rc = c->d_fredo.ensure((Tn + 4) * 4)) || (rc = c->d_task_lists.ensure(2 * Tn * 4))) return rc;
	f.tasks = c->d_tasks.as<int2>(); f.alns = c->d_alns.as<DevAln>(); f.redo_list = c->d_fredo.as<int>();
	f.fast_list = c->d_task_lists.as<int>(); f.dp_list = f.fast_list + Tn; f.list_n = (int*)(pool_head + 15);
	if ((rc = launch_task_fill(f, c->stream))) return rc;
	int n_list[2] = { 0, 0 };
	HIP_TRY(hipMemcpyAsync(n_list, f.list_n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	size_t want_pool = std::max(c->pool_cap, Tn * 64 + Tn * 32 + (size_t)(16 << 20));   // 64-byte slot per task + a shared tail for long texts
	for (int attempt = 0;; ++attempt) {
		if ((rc = c->d_pool.ensure(want_pool))) return rc;
		c->pool_cap = want_pool;
		f.pool = c->d_pool.as<uint8_t>(); f.pool_cap = want_pool;
		HIP_TRY(hipMemsetAsync(c->d_fmisc.p, 0, 120, c->stream));   // pool head, redo count, error; the task-list lengths at +120 stay
		{ const unsigned long long head0 = (unsigned long long)Tn * 64; HIP_TRY(hipMemcpyAsync(c->d_fmisc.p, &head0, 8, hipMemcpyHostToDevice, c->stream)); }   // the shared tail starts behind the slots
		if (n_list[1] > 0) { if ((rc = c->d_zslab.ensure(cigar_zslab_bytes(c->max_len, n_list[1])))) return rc; f.zslab = c->d_zslab.as<unsigned>(); }
		if ((rc = launch_cigar(f, n_list[0], n_list[1], c->max_len, c->stream, c->stream2, c->ev_fork, c->ev_join))) return rc;
		int h[6] = { 0 };
		HIP_TRY(hipMemcpyAsync(h, c->d_fmisc.p, 24, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		int n_redo = h[2], err = h[4];
		if (err == 5 && attempt < 6) { want_pool *= 4; continue; }     // text pool exhausted: larger pool, run the stage again (GPU only)
		if (!err && n_redo > 0) {                                       // tasks beyond the LDS variant: global-slab variant
			const int grid = std::min(n_redo, 256);                   // one 5.7 MB slab per workgroup
			if ((rc = c->d_bigz.ensure((size_t)grid * cigar_big_slab_bytes()))) return rc;
			f.big_z = c->d_bigz.as<uint8_t>();
			if ((rc = launch_cigar_big(f, grid, c->stream))) return rc;
			HIP_TRY(hipMemcpyAsync(h, c->d_fmisc.p, 24, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			err = h[4];
			if (err == 5 && attempt < 6) { want_pool *= 4; continue; }
		}
		if (err) {
			fprintf(stderr, "[bwahip] alignment kernel reported code %d (read %d of the batch)%s\n", err, h[5], err == 6 ? ": a region's reference span / band exceeds the compiled limits" : "");
			return err == 6 ? BWAHIP_ECAPACITY : BWAHIP_EINTERNAL;
		}
		break;
	}
	if (timed) HIP_TRY(hipEventRecord(c->ev[17], c->stream));
	if ((rc = pe ? launch_sam_pe(f, false, c->stream) : launch_sam(f, false, c->stream))) return rc;
	if ((rc = launch_scan(c->d_sam_len.as<int>(), c->d_sam_off.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	if (timed) HIP_TRY(hipEventRecord(c->ev[18], c->stream));
	int64_t total = 0;
	HIP_TRY(hipMemcpyAsync(&total, c->d_sam_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->total_sam = total;
	if ((rc = c->d_sam.ensure((size_t)total + 64))) return rc;
	f.sam = c->d_sam.as<uint8_t>();
	if (c->want_host_sam_off) {                                 // bwahip_process_seqs: the offsets travel ahead of the write pass
		c->h_sam_off.resize((size_t)n + 1);
		HIP_TRY(hipMemcpyAsync(c->h_sam_off.data(), c->d_sam_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	}
	// the write pass in two halves (cut at an even read: mates stay together) with an event between them: a caller that downloads the text
	// (bwahip_process_seqs) starts on the first half while the second is being written
	c->sam_half_reads = (n / 2) & ~1;
	if ((rc = pe ? launch_sam_pe(f, true, c->stream, 0, c->sam_half_reads) : launch_sam(f, true, c->stream, 0, c->sam_half_reads))) return rc;
	HIP_TRY(hipEventRecord(c->ev_sam_half, c->stream));
	if ((rc = pe ? launch_sam_pe(f, true, c->stream, c->sam_half_reads, n) : launch_sam(f, true, c->stream, c->sam_half_reads, n))) return rc;
	if (timed) {
		HIP_TRY(hipEventRecord(c->ev[19], c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[0], c->ev[15], c->ev[16]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[1], c->ev[16], c->ev[17]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[2], c->ev[17], c->ev[18]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[3], c->ev[18], c->ev[19]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[4], c->ev[20], c->ev[15]));   // paired-end: insert sizes + mate rescue
	}
	return 0;
}

#define NT4_TABLE \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,5,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4, \
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4
static const uint8_t k_nt4[256] = { NT4_TABLE };      // nst_nt4_table, bntseq.c:46
__constant__ uint8_t d_nt4[256] = { NT4_TABLE };

// bwamem.c:1067-1068 on the batch in HBM: seq[i] = seq[i] < 4 ? seq[i] : nst_nt4_table[seq[i]] (seq is char: bytes >= 128 compare
// below 4 and stay as they are, as in the reference), 16 bases per thread
__global__ __launch_bounds__(256) void k_nt4_conv(uint8_t *seq, int64_t n)
{
	const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
	if (i >= n) return;
	if (i + 16 <= n) {
		uint4 v = *reinterpret_cast<const uint4*>(seq + i);
		uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			uint32_t o = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) { const uint32_t c = w[k] >> (8 * b) & 0xff; o |= (uint32_t)(((int8_t)c < 4) ? c : d_nt4[c]) << (8 * b); }
			w[k] = o;
		}
		*reinterpret_cast<uint4*>(seq + i) = make_uint4(w[0], w[1], w[2], w[3]);
	} else for (int64_t k = i; k < n; ++k) { const uint8_t c = seq[k]; seq[k] = ((int8_t)c < 4) ? c : d_nt4[c]; }
}
int launch_nt4(uint8_t *seq, int64_t n, hipStream_t st)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_nt4_conv, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, st, seq, n);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

// One batch of host reads on its way to HBM.  The base codes go first (converted in place exactly as bwamem.c:1067-1068 does)
// and the hot path starts on them at once; qualities, names and comments -- needed only when the SAM text is written -- are
// gathered and uploaded by a helper thread on the copy stream while the hot path runs.
static int stage_codes(bwahip_ctx *c, int nt, int n, bwahip_seq_t *seqs, BatchText &t)
{
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	t.off.resize(n + 1); t.qoff.resize(n); t.noff.resize(n + 1); t.coff.resize(n + 1);
	t.off[0] = t.noff[0] = t.coff[0] = 0;
	// offsets of bases / qualities / names / comments: per-chunk totals in parallel (two strlen per read), a serial scan over the
	// chunks, then the running sums inside every chunk in parallel
	const int C = n >= 4096 && nt > 1 ? nt : 1;
	struct Tot { int64_t seq = 0, qual = 0, name = 0, comm = 0; int bad = 0, any_comm = 0; char pad[24]; };
	std::vector<Tot> tot(C + 1);
	auto chunk = [&](int c) { return (int64_t)n * c / C; };
	auto for_chunks = [&](auto f) {
		if (C == 1) { f(0); return; }
		std::vector<std::thread> th;
		for (int c = 0; c < C; ++c) th.emplace_back(f, c);
		for (auto &x : th) x.join();
	};
	for_chunks([&](int c) {
		Tot x;
		for (int64_t i = chunk(c); i < chunk(c + 1); ++i) {
			if (seqs[i].l_seq < 0 || !seqs[i].name) { x.bad = 1; continue; }
			const int64_t ln = (int64_t)strlen(seqs[i].name) + 1, lc = seqs[i].comment ? (int64_t)strlen(seqs[i].comment) : 0;
			t.noff[i + 1] = ln; t.coff[i + 1] = lc ? lc + 1 : 0;
			x.seq += seqs[i].l_seq; x.qual += seqs[i].qual ? seqs[i].l_seq : 0; x.name += ln; x.comm += lc ? lc + 1 : 0; x.any_comm |= lc > 0;
		}
		tot[c + 1] = x;
	});
	for (int c = 1; c <= C; ++c) {
		if (tot[c].bad) return BWAHIP_EINVAL;
		t.any_comment |= tot[c].any_comm != 0;
		tot[c].seq += tot[c - 1].seq; tot[c].qual += tot[c - 1].qual; tot[c].name += tot[c - 1].name; tot[c].comm += tot[c - 1].comm;
	}
	t.qtot = tot[C].qual;
	for_chunks([&](int c) {
		int64_t so = tot[c].seq, qo = tot[c].qual, no = tot[c].name, co = tot[c].comm;
		for (int64_t i = chunk(c); i < chunk(c + 1); ++i) {
			so += seqs[i].l_seq; t.off[i + 1] = so;
			if (seqs[i].qual) { t.qoff[i] = qo; qo += seqs[i].l_seq; } else t.qoff[i] = -1;
			no += t.noff[i + 1]; t.noff[i + 1] = no;
			co += t.coff[i + 1]; t.coff[i + 1] = co;
		}
	});
	const double t1 = now();
	// one pinned staging buffer (kept by the context) holds codes | qualities | names | comments: the copies to HBM then run at
	// PCIe speed instead of through pageable memory
	t.sz_codes = ((size_t)t.off[n] + 64) & ~(size_t)63; t.sz_qual = ((size_t)t.qtot + 127) & ~(size_t)63; t.sz_names = ((size_t)t.noff[n] + 127) & ~(size_t)63;
	t.sz_comm = t.any_comment ? ((size_t)t.coff[n] + 127) & ~(size_t)63 : 0;
	int rc = c->h_stage.ensure(t.sz_codes + t.sz_qual + t.sz_names + t.sz_comm);
	if (rc) return rc;
	// device buffers of the text are sized here, on the calling thread, so that the helper only copies
	if ((rc = c->d_qual.ensure(t.sz_qual)) || (rc = c->d_qual_off.ensure((size_t)n * 8 + 16)) || (rc = c->d_names.ensure(t.sz_names)) ||
	    (rc = c->d_name_off.ensure((size_t)(n + 1) * 8)) || (rc = c->d_comment_off.ensure((size_t)(n + 1) * 8))) return rc;
	if (t.any_comment) { if ((rc = c->d_comments.ensure(t.sz_comm))) return rc; }
	else c->d_comments.release();
	// the bases travel as the caller wrote them (ASCII or codes) and are turned into codes in HBM (k_nt4: the same table look-up as
	// bwamem.c:1067-1068); the caller's own arrays are converted in place by the helper thread while the hot path runs
	const double t2 = now();
	uint8_t *codes = (uint8_t*)c->h_stage.p;
	par_for_chunks(n, nt, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) memcpy(codes + t.off[i], seqs[i].seq, (size_t)seqs[i].l_seq); });
	const double t3 = now();
	if ((rc = bwahip_batch_upload(c, n, codes, t.off.data()))) return rc;
	if (c->knobs.e2e_log) fprintf(stderr, "[bwahip] stage_codes: offsets %.1f ms, buffers %.1f ms, gather %.1f ms, upload %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now() - t3) * 1e3);
	return launch_nt4(c->d_seq.as<uint8_t>(), c->total_bases, c->stream);
}

// helper thread: the text of the batch into the pinned buffer and on to HBM (copy stream); returns when the copies are done
static int stage_text(bwahip_ctx *c, int nt, int n, bwahip_seq_t *seqs, const BatchText &t)
{
	HIP_TRY(hipSetDevice(c->device));
	uint8_t *qual = (uint8_t*)c->h_stage.p + t.sz_codes, *names = qual + t.sz_qual, *comments = names + t.sz_names;
	memset(qual + (size_t)t.qtot, 0, t.sz_qual - (size_t)t.qtot); memset(names + (size_t)t.noff[n], 0, t.sz_names - (size_t)t.noff[n]);
	if (t.any_comment) memset(comments + (size_t)t.coff[n], 0, t.sz_comm - (size_t)t.coff[n]);
	par_for_chunks(n, nt, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			char *s = seqs[i].seq;                                  // bwamem.c:1067-1068, in place (the reference's contract)
			for (int k = 0; k < seqs[i].l_seq; ++k) s[k] = s[k] < 4 ? s[k] : (char)k_nt4[(uint8_t)s[k]];
			if (t.qoff[i] >= 0) memcpy(&qual[t.qoff[i]], seqs[i].qual, seqs[i].l_seq);
			memcpy(&names[t.noff[i]], seqs[i].name, t.noff[i + 1] - t.noff[i]);
			if (t.any_comment && t.coff[i + 1] > t.coff[i]) memcpy(&comments[t.coff[i]], seqs[i].comment, t.coff[i + 1] - t.coff[i]);
		}
	});
	hipStream_t st = c->stream_copy;
	HIP_TRY(hipMemcpyAsync(c->d_qual.p, qual, t.sz_qual, hipMemcpyHostToDevice, st));
	if (n) HIP_TRY(hipMemcpyAsync(c->d_qual_off.p, t.qoff.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(c->d_names.p, names, t.sz_names, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(c->d_name_off.p, t.noff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(hipMemcpyAsync(c->d_comment_off.p, t.coff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
	if (t.any_comment) HIP_TRY(hipMemcpyAsync(c->d_comments.p, comments, t.sz_comm, hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	return 0;
}

// text != nullptr: the batch's SAM stays one piece (in the context's pinned buffer) instead of being cut into per-read strings
static int process_seqs_impl(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0,
                             const char **text_out, int64_t *len_out, const int64_t **off_out)
{
	if (!ctx || !opt || n < 0 || (n && !seqs)) return BWAHIP_EINVAL;
	const bool pe = (opt->flag & BWAHIP_F_PE) != 0;
	if (pe && (n & 1)) return BWAHIP_EINVAL;
	if (text_out) { *text_out = ""; *len_out = 0; if (off_out) *off_out = nullptr; }
	if (!ctx->knobs.gpu_final || (pe && !ctx->knobs.gpu_pair)) {
		if (text_out) return BWAHIP_EINVAL;                       // the one-piece output exists on the GPU path only
		host_final_fn hf = load_host_final();
		return hf ? hf(ctx, opt, n_processed, n, seqs, pes0) : BWAHIP_EINVAL;
	}
	if (pe) for (int i = 0; i < n; i += 2) if (strcmp(seqs[i].name, seqs[i + 1].name) != 0) { fprintf(stderr, "[bwahip] paired reads have different names\n"); return BWAHIP_EINVAL; }   // err_fatal in the reference (bwamem_pair.c:386)
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(ctx->device));
	const bool verbose = ctx->knobs.verbose != 0;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	const int nt = opt->n_threads > 1 ? opt->n_threads : 1;
	BatchText &bt = ctx->batch_text;                             // kept by the context: no fresh page faults per batch
	bt.qtot = 0; bt.any_comment = false;
	int rc = stage_codes(ctx, nt, n, seqs, bt);
	if (rc) return rc;
	const double t1 = now();
	// the text travels while the hot path runs (half of the host threads: the other half of the machine belongs to the caller's reader)
	int rc_text = 0;
	std::thread text_thread([&] { rc_text = stage_text(ctx, nt > 2 ? nt / 2 : 1, n, seqs, bt); });
	rc = run_pipeline(ctx, opt, false, false);
	text_thread.join();
	if (rc || (rc = rc_text)) return rc;
	const double t2 = now();
	ctx->want_host_sam_off = true;
	rc = run_final(ctx, opt, n_processed, pes0, false);
	ctx->want_host_sam_off = false;
	if (rc) return rc;
	// SAM text back through the pinned buffer, on the context's stream (a non-blocking stream: a plain hipMemcpy would not wait
	// for the SAM kernel), in slices of reads: while slice k+1 travels, the host threads cut slice k into one malloc()ed string
	// per read, which is what the reference's contract wants (bwamem.c:1054)
	// SAM text back through the pinned buffer.  run_final wrote it in two halves and sent the offsets ahead: once the first half is written
	// (ev_sam_half) its download starts on the copy stream, beside the kernel that writes the second half.  Per-read strings: in slices of reads --
	// while slice k+1 travels, the host threads cut slice k into one malloc()ed string per read, which is what the reference's contract wants
	// (bwamem.c:1054).  (The context's streams are non-blocking: a plain hipMemcpy would not wait for the SAM kernels.)
	std::vector<int64_t> &soff = ctx->h_sam_off;
	HostBuf &hs = text_out && (ctx->sam_flip ^= 1) ? ctx->h_sam2 : ctx->h_sam;   // one-piece callers: the text stays valid until the next-but-one call
	if ((rc = hs.ensure((size_t)ctx->total_sam + 1))) return rc;
	char *text = (char*)hs.p;
	HIP_TRY(hipEventSynchronize(ctx->ev_sam_half));               // first half written; the offsets arrived before that
	const int half = ctx->sam_half_reads;
	constexpr int SLICES = 8;
	int64_t sl_beg[SLICES + 1];
	for (int k = 0; k <= SLICES / 2; ++k) sl_beg[k] = (int64_t)half * k / (SLICES / 2);
	for (int k = 0; k <= SLICES / 2; ++k) sl_beg[SLICES / 2 + k] = half + (int64_t)(n - half) * k / (SLICES / 2);
	auto copy_slice = [&](int k, hipStream_t st) -> int {
		const int64_t b = soff[sl_beg[k]], e = soff[sl_beg[k + 1]];
		if (e > b) HIP_TRY(hipMemcpyAsync(text + b, (const char*)ctx->d_sam.p + b, (size_t)(e - b), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipEventRecord(ctx->ev_slice[k], st));
		return 0;
	};
	for (int k = 0; k < SLICES / 2; ++k) if ((rc = copy_slice(k, ctx->stream_copy))) return rc;
	for (int k = SLICES / 2; k < SLICES; ++k) if ((rc = copy_slice(k, ctx->stream))) return rc;
	const int n_sl = SLICES;
	if (text_out) {                                               // one piece: no per-read strings
		HIP_TRY(hipStreamSynchronize(ctx->stream_copy));
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		const double t4 = now();
		text[ctx->total_sam] = 0;
		*text_out = text; *len_out = ctx->total_sam; if (off_out) *off_out = soff.data();
		par_for_chunks(n, nt, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) seqs[i].sam = nullptr; });
		if (verbose || ctx->knobs.e2e_log)
			fprintf(stderr, "[bwahip] process_seqs_text %d reads: codes gather+upload %.1f ms, hot path (text upload beside it) %.1f ms, finalisation+SAM on GPU and download %.1f ms (%lld bytes), rest %.1f ms\n",
			        n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t2) * 1e3, (long long)ctx->total_sam, (now() - t4) * 1e3);
		return 0;
	}
	const double t3 = now();
	std::atomic<int> oom(0), hip_bad(0);
	auto cut = [&](int tid, int nthr) {
		(void)hipSetDevice(ctx->device);
		for (int k = 0; k < n_sl; ++k) {
			if (hipEventSynchronize(ctx->ev_slice[k]) != hipSuccess) { hip_bad = 1; return; }
			const int64_t cnt = sl_beg[k + 1] - sl_beg[k], per = (cnt + nthr - 1) / nthr;
			const int64_t b = sl_beg[k] + tid * per, e = b + per < sl_beg[k + 1] ? b + per : sl_beg[k + 1];
			for (int64_t i = b; i < e; ++i) {
				const size_t len = (size_t)(soff[i + 1] - soff[i]);
				char *p = (char*)malloc(len + 1);
				if (!p) { oom = 1; seqs[i].sam = nullptr; continue; }
				memcpy(p, text + soff[i], len); p[len] = 0;
				seqs[i].sam = p;
			}
		}
	};
	if (nt == 1) cut(0, 1);
	else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(cut, t, nt); for (auto &x : th) x.join(); }
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if (verbose || ctx->knobs.e2e_log)
		fprintf(stderr, "[bwahip] process_seqs %d reads: codes gather+upload %.1f ms, hot path (text upload beside it) %.1f ms, finalisation+SAM on GPU %.1f ms (%lld bytes), download + per-read strings %.1f ms\n",
		        n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (long long)ctx->total_sam, (now() - t3) * 1e3);
	if (hip_bad) return BWAHIP_ENODEV;
	return oom ? BWAHIP_ENOMEM : 0;
}

extern "C" int bwahip_process_seqs(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0)
{
	return process_seqs_impl(ctx, opt, n_processed, n, seqs, pes0, nullptr, nullptr, nullptr);
}

// The same work with the batch's SAM handed over in one piece -- for a caller whose output step is one fwrite (fastmap.c prints
// seqs[i].sam read by read): no malloc per read, no second copy.  *sam: NUL-terminated text of the whole batch in read order, *off
// (optional): n + 1 offsets, read i's records are sam[off[i] .. off[i+1]).  Both live in the context and stay valid until its next call.
extern "C" int bwahip_process_seqs_text(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0,
                                        const char **sam, int64_t *sam_len, const int64_t **off)
{
	if (!sam || !sam_len) return BWAHIP_EINVAL;
	return process_seqs_impl(ctx, opt, n_processed, n, seqs, pes0, sam, sam_len, off);
}

// Insert-size statistics (mem_pestat_t x 4: FF, FR, RF, RR) and mate-rescue counters ([0] Smith-Waterman alignments run on
// the GPU, [1] regions they added, [2] most alignments of one pair, [3] pairs that needed any) of the last paired-end batch
// finalised on the GPU.
extern "C" int bwahip_last_pe_stats(bwahip_ctx *ctx, bwahip_pestat_t *pes4, uint64_t *counters2)
{
	if (!ctx) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	if (pes4) memcpy(pes4, ctx->last_pes, sizeof ctx->last_pes);
	if (counters2) for (int i = 0; i < 4; ++i) counters2[i] = ctx->last_pe_counters[i];
	if (getenv("BWAHIP_PE_LOG")) {
		const unsigned long long *c = ctx->last_pe_counters;
		fprintf(stderr, "[bwahip] mate rescue: %llu alignments run ahead by k_matesw_sw; mem_matesw used %llu (%llu more inside k_matesw), %llu regions added, %llu pairs; k_matesw summed over its wavefronts: window fetch %.1f ms, alignments %.1f ms, list clean-up %.1f ms; longest pair %.2f ms\n",
		        ctx->last_sw_tasks, c[0], c[8], c[1], c[3], c[4] / 1e5, c[5] / 1e5, c[6] / 1e5, c[7] / 1e5);
	}
	return 0;
}
