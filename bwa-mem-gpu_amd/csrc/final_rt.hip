// Host sequencing of the GPU finalisation (regions of mem_align1_core in HBM -> SAM text in HBM) and the C ABI entry
// bwahip_process_seqs == mem_process_seqs (bwamem.h:69).  Kernels: k_final.hip (mark primary, selection, CIGAR/NM/MD/mapQ),
// k_sam.hip (SAM text).  Paired-end batches still take the host path of host_final.cpp after the GPU hot path.
#include "ctx_internal.h"
#include <atomic>
#include <thread>
#include <math.h>
#include <algorithm>
#include <chrono>

int bwahip_process_seqs_host(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0);   // host_final.cpp

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
	    (rc = c->d_fmisc.ensure(128))) return rc;
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
static int run_pe_rescue(bwahip_ctx *c, const bwahip_opt_t *opt, const DevOpt &dopt, int64_t n_processed, const bwahip_pestat_t *pes0, PairLaunch &pl)
{
	const int n = c->n_reads;
	int rc;
	memset(&pl, 0, sizeof pl);
	pl.ix = c->ix; pl.opt = dopt; pl.n_reads = n; pl.seq = c->d_seq.as<uint8_t>(); pl.off = c->d_off.as<int64_t>(); pl.n_processed = n_processed;
	pl.logtab = c->d_logtab.as<double>();
	pl.regs = c->d_regs.as<DevReg>(); pl.reg_base = c->d_reg_base.as<int64_t>(); pl.reg_n = c->d_reg_n.as<int>();
	unsigned long long *fm = c->d_fmisc.as<unsigned long long>();
	pl.err = (int*)(fm + 2); pl.resc_n = (int*)(fm + 4); pl.counters = fm + 5; pl.sw_n = (int*)(fm + 14);
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
	    (rc = c->d_resc.ensure((size_t)(n / 2 + 4) * 4)) || (rc = c->d_sw_cnt.ensure((size_t)n * 4)) || (rc = c->d_sw_base.ensure((size_t)(n + 1) * 8))) return rc;
	pl.sw_cnt = c->d_sw_cnt.as<int>(); pl.sw_base = c->d_sw_base.as<int64_t>();
	pl.nb = c->d_nb.as<int>(); pl.pe_cap = c->d_pe_cap.as<int>(); pl.pe_base = c->d_pe_base.as<int64_t>(); pl.pe_n = c->d_pe_n.as<int>(); pl.resc_list = c->d_resc.as<int>();
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
		pl.sw_res = c->d_sw_res.as<SwRes>(); pl.sw_tasks = c->d_sw_tasks.as<int>(); pl.sw_info = c->d_sw_info.as<int2>();
	}
	const size_t R = (size_t)(cap ? cap : 1);
	c->total_regs = cap;                                          // from here on the region slots are the paired-end ones
	if ((rc = c->d_pe_regs.ensure(R * sizeof(DevReg))) || (rc = c->d_pe_tmp.ensure(R * sizeof(DevReg))) || (rc = c->d_pe_keys.ensure(R * 16)) || (rc = c->d_pe_idx.ensure(R * 8))) return rc;
	pl.pe_regs = c->d_pe_regs.as<DevReg>(); pl.pe_tmp = c->d_pe_tmp.as<DevReg>(); pl.pe_keys = c->d_pe_keys.p; pl.pe_idx = c->d_pe_idx.as<int>();
	if ((rc = launch_pe_copy(pl, c->stream))) return rc;
	int n_resc = 0, n_sw_tasks = 0;
	HIP_TRY(hipMemcpyAsync(&n_resc, pl.resc_n, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(&n_sw_tasks, pl.sw_n, 4, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if ((rc = launch_matesw_sw(pl, n_sw_tasks, c->stream))) return rc;   // the alignments against the unrescued lists, all at once
	if (n_resc > 0) {
		const int grid = std::min(n_resc, 2048);
		pl.slab_stride = (matesw_slab_bytes(widest + c->max_len) + 255) & ~(size_t)255;
		if ((rc = c->d_ms_slab.ensure(pl.slab_stride * (size_t)grid))) return rc;
		pl.slab = c->d_ms_slab.as<uint8_t>();
		if ((rc = launch_matesw(pl, grid, c->stream))) return rc;
	}
	HIP_TRY(hipMemcpyAsync(c->last_pe_counters, pl.counters, 72, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->last_pe_counters[3] = (unsigned long long)n_resc;
	if (c->knobs.verbose) fprintf(stderr, "[bwahip] mate rescue: %llu SW, %llu added, max %llu per pair, %d pairs, %llu aligned inside the sequential pass; ticks(10ns) window %llu sw %llu dedup %llu, slowest pair %llu\n", c->last_pe_counters[0], c->last_pe_counters[1], c->last_pe_counters[2], n_resc, c->last_pe_counters[8], c->last_pe_counters[4], c->last_pe_counters[5], c->last_pe_counters[6], c->last_pe_counters[7]);
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
	HIP_TRY(hipMemsetAsync(c->d_fmisc.p, 0, 128, c->stream));
	if (timed) HIP_TRY(hipEventRecord(c->ev[20], c->stream));
	PairLaunch pl;
	const DevOpt dopt_pe = make_dev_opt(opt);
	if (pe && (rc = run_pe_rescue(c, opt, dopt_pe, n_processed, pes0, pl))) return rc;
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
	if ((rc = launch_mark_primary(f, !pe, c->stream))) return rc;
	if (pe) {                                                     // mem_pair + the decisions of mem_sam_pe
		if ((rc = c->d_pe_read.ensure((size_t)n * sizeof(PeRead)))) return rc;
		pl.fregs = f.fregs; pl.fregs_w = f.fregs; pl.freg_n = f.freg_n; pl.n_pri = f.n_pri; pl.need = f.need; pl.xa_owner = f.xa_owner;
		pl.task_n = f.task_n; pl.rec_n = f.rec_n; pl.scr = f.scr; pl.pe_read = c->d_pe_read.as<PeRead>();
		if ((rc = launch_pair(pl, c->stream))) return rc;
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
		if ((rc = launch_cigar(f, n_list[0], n_list[1], c->max_len, c->stream, c->stream2, c->ev_fork, c->ev_join))) return rc;
		int h[6] = { 0 };
		HIP_TRY(hipMemcpyAsync(h, c->d_fmisc.p, 24, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		int n_redo = h[2], err = h[4];
		if (err == 5 && attempt < 6) { want_pool *= 4; continue; }     // text pool exhausted: larger pool, run the stage again (GPU only)
		if (!err && n_redo > 0) {                                       // tasks beyond the LDS variant: global-slab variant
			const int grid = std::min(n_redo, 32);
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
	if ((rc = pe ? launch_sam_pe(f, true, c->stream) : launch_sam(f, true, c->stream))) return rc;
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

static const uint8_t k_nt4[256] = {      // nst_nt4_table, bntseq.c:46
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,5,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,
	4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4,  4,4,4,4, 4,4,4,4, 4,4,4,4, 4,4,4,4
};

// Upload one batch of host reads: base codes (converted in place exactly as bwamem.c:1067-1068 does), qualities, names, comments.
static int upload_batch_text(bwahip_ctx *c, int nt, int n, bwahip_seq_t *seqs)
{
	std::vector<int64_t> off(n + 1, 0), qoff(n, -1), noff(n + 1, 0), coff(n + 1, 0);
	int64_t qtot = 0;
	bool any_comment = false;
	for (int i = 0; i < n; ++i) {
		if (seqs[i].l_seq < 0 || !seqs[i].name) return BWAHIP_EINVAL;
		off[i + 1] = off[i] + seqs[i].l_seq;
		if (seqs[i].qual) { qoff[i] = qtot; qtot += seqs[i].l_seq; }
		noff[i + 1] = noff[i] + (int64_t)strlen(seqs[i].name) + 1;
		const int64_t lc = seqs[i].comment ? (int64_t)strlen(seqs[i].comment) : 0;
		coff[i + 1] = coff[i] + (lc ? lc + 1 : 0);
		any_comment |= lc > 0;
	}
	// one pinned staging buffer (kept by the context) holds codes | qualities | names | comments: the copies to HBM then run at
	// PCIe speed instead of through pageable memory
	const size_t sz_codes = ((size_t)off[n] + 64) & ~(size_t)63, sz_qual = ((size_t)qtot + 127) & ~(size_t)63, sz_names = ((size_t)noff[n] + 127) & ~(size_t)63,
	             sz_comm = any_comment ? ((size_t)coff[n] + 127) & ~(size_t)63 : 0;
	int rc = c->h_stage.ensure(sz_codes + sz_qual + sz_names + sz_comm);
	if (rc) return rc;
	uint8_t *codes = (uint8_t*)c->h_stage.p, *qual = codes + sz_codes, *names = qual + sz_qual, *comments = names + sz_names;
	memset(qual + (size_t)qtot, 0, sz_qual - (size_t)qtot); memset(names + (size_t)noff[n], 0, sz_names - (size_t)noff[n]);
	if (any_comment) memset(comments + (size_t)coff[n], 0, sz_comm - (size_t)coff[n]);
	par_for_chunks(n, nt, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			char *s = seqs[i].seq;
			for (int k = 0; k < seqs[i].l_seq; ++k) { s[k] = s[k] < 4 ? s[k] : (char)k_nt4[(uint8_t)s[k]]; codes[off[i] + k] = (uint8_t)s[k]; }
			if (qoff[i] >= 0) memcpy(&qual[qoff[i]], seqs[i].qual, seqs[i].l_seq);
			memcpy(&names[noff[i]], seqs[i].name, noff[i + 1] - noff[i]);
			if (any_comment && coff[i + 1] > coff[i]) memcpy(&comments[coff[i]], seqs[i].comment, coff[i + 1] - coff[i]);
		}
	});
	rc = bwahip_batch_upload(c, n, codes, off.data());
	if (rc) return rc;
	if ((rc = dev_upload(c->d_qual, qual, sz_qual, c->stream)) || (rc = dev_upload(c->d_qual_off, qoff.data(), (size_t)n * 8, c->stream)) ||
	    (rc = dev_upload(c->d_names, names, sz_names, c->stream)) || (rc = dev_upload(c->d_name_off, noff.data(), (size_t)(n + 1) * 8, c->stream)) ||
	    (rc = dev_upload(c->d_comment_off, coff.data(), (size_t)(n + 1) * 8, c->stream))) return rc;
	if (any_comment) { if ((rc = dev_upload(c->d_comments, comments, sz_comm, c->stream))) return rc; }
	else c->d_comments.release();
	HIP_TRY(hipStreamSynchronize(c->stream));
	return 0;
}

extern "C" int bwahip_process_seqs(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0)
{
	if (!ctx || !opt || n < 0 || (n && !seqs)) return BWAHIP_EINVAL;
	const bool pe = (opt->flag & BWAHIP_F_PE) != 0;
	if (pe && (n & 1)) return BWAHIP_EINVAL;
	if (!ctx->knobs.gpu_final || (pe && !ctx->knobs.gpu_pair)) return bwahip_process_seqs_host(ctx, opt, n_processed, n, seqs, pes0);
	if (pe) for (int i = 0; i < n; i += 2) if (strcmp(seqs[i].name, seqs[i + 1].name) != 0) { fprintf(stderr, "[bwahip] paired reads have different names\n"); return BWAHIP_EINVAL; }   // err_fatal in the reference (bwamem_pair.c:386)
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(ctx->device));
	const bool verbose = ctx->knobs.verbose != 0;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	int rc = upload_batch_text(ctx, opt->n_threads, n, seqs);
	if (rc) return rc;
	const double t1 = now();
	if ((rc = run_pipeline(ctx, opt, false, false))) return rc;
	const double t2 = now();
	if ((rc = run_final(ctx, opt, n_processed, pes0, false))) return rc;
	// SAM text back in one piece (pinned staging buffer, on the context's stream: it is a non-blocking stream, a plain hipMemcpy
	// would not wait for the SAM kernel), then one malloc()ed string per read as the reference's contract wants (bwamem.c:1054)
	std::vector<int64_t> soff(n + 1);
	if ((rc = ctx->h_sam.ensure((size_t)ctx->total_sam + 1))) return rc;
	char *text = (char*)ctx->h_sam.p;
	HIP_TRY(hipMemcpyAsync(soff.data(), ctx->d_sam_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (ctx->total_sam) HIP_TRY(hipMemcpyAsync(text, ctx->d_sam.p, (size_t)ctx->total_sam, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	const double t3 = now();
	std::atomic<int> oom(0);
	par_for_chunks(n, opt->n_threads, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			const size_t len = (size_t)(soff[i + 1] - soff[i]);
			char *p = (char*)malloc(len + 1);
			if (!p) { oom = 1; seqs[i].sam = nullptr; continue; }
			memcpy(p, text + soff[i], len); p[len] = 0;
			seqs[i].sam = p;
		}
	});
	if (verbose) fprintf(stderr, "[bwahip] process_seqs %d reads: gather+upload %.1f ms, hot path %.1f ms, finalisation+SAM on GPU+download %.1f ms (%lld bytes), per-read strings %.1f ms\n",
	                     n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (long long)ctx->total_sam, (now() - t3) * 1e3);
	return oom ? BWAHIP_ENOMEM : 0;
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
	return 0;
}
