// Host sequencing of the GPU finalisation (regions of mem_align1_core in HBM -> SAM text in HBM) and the C ABI entry
// bwahip_process_seqs == mem_process_seqs (bwamem.h:69).  Kernels: k_final.hip (mark primary, selection, CIGAR/NM/MD/mapQ),
// k_sam.hip (SAM text).  Paired-end batches still take the host path of host_final.cpp after the GPU hot path.
#include "ctx_internal.h"
#include <atomic>
#include <thread>

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
	    (rc = c->d_fmisc.ensure(64))) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return 0;
}

// K6 -> K9 over the batch run_pipeline left in HBM.  The text inputs (d_qual, d_names, ...) must be uploaded.
int run_final_se(bwahip_ctx *c, const bwahip_opt_t *opt, int64_t n_processed, bool timed)
{
	const int n = c->n_reads;
	c->total_sam = 0; c->total_tasks = 0;
	if (n == 0) return 0;
	const size_t R = (size_t)(c->total_regs ? c->total_regs : 1);
	int rc;
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
	HIP_TRY(hipMemsetAsync(c->d_fmisc.p, 0, 64, c->stream));
	if (timed) HIP_TRY(hipEventRecord(c->ev[15], c->stream));
	if ((rc = launch_mark_primary(f, true, c->stream))) return rc;
	if ((rc = launch_scan(c->d_task_n.as<int>(), c->d_task_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	if (timed) HIP_TRY(hipEventRecord(c->ev[16], c->stream));
	int64_t T = 0;
	HIP_TRY(hipMemcpyAsync(&T, c->d_task_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->total_tasks = T;
	const size_t Tn = (size_t)(T ? T : 1);
	if ((rc = c->d_tasks.ensure(Tn * 8)) || (rc = c->d_alns.ensure(Tn * sizeof(DevAln))) || (rc = c->d_fredo.ensure((Tn + 4) * 4))) return rc;
	f.tasks = c->d_tasks.as<int2>(); f.alns = c->d_alns.as<DevAln>(); f.redo_list = c->d_fredo.as<int>();
	if ((rc = launch_task_fill(f, c->stream))) return rc;
	size_t want_pool = std::max(c->pool_cap, Tn * 96 + (size_t)(16 << 20));
	for (int attempt = 0;; ++attempt) {
		if ((rc = c->d_pool.ensure(want_pool))) return rc;
		c->pool_cap = want_pool;
		f.pool = c->d_pool.as<uint8_t>(); f.pool_cap = want_pool;
		HIP_TRY(hipMemsetAsync(c->d_fmisc.p, 0, 64, c->stream));
		if ((rc = launch_cigar(f, T, c->stream))) return rc;
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
	if ((rc = launch_sam(f, false, c->stream))) return rc;
	if ((rc = launch_scan(c->d_sam_len.as<int>(), c->d_sam_off.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
	if (timed) HIP_TRY(hipEventRecord(c->ev[18], c->stream));
	int64_t total = 0;
	HIP_TRY(hipMemcpyAsync(&total, c->d_sam_off.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->total_sam = total;
	if ((rc = c->d_sam.ensure((size_t)total + 64))) return rc;
	f.sam = c->d_sam.as<uint8_t>();
	if ((rc = launch_sam(f, true, c->stream))) return rc;
	if (timed) {
		HIP_TRY(hipEventRecord(c->ev[19], c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[0], c->ev[15], c->ev[16]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[1], c->ev[16], c->ev[17]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[2], c->ev[17], c->ev[18]));
		HIP_TRY(hipEventElapsedTime(&c->final_ms[3], c->ev[18], c->ev[19]));
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
	std::vector<uint8_t> codes((size_t)off[n] + 1), qual((size_t)qtot + 64), names((size_t)noff[n] + 64, 0), comments(any_comment ? (size_t)coff[n] + 64 : 0, 0);
	par_for_chunks(n, nt, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			char *s = seqs[i].seq;
			for (int k = 0; k < seqs[i].l_seq; ++k) { s[k] = s[k] < 4 ? s[k] : (char)k_nt4[(uint8_t)s[k]]; codes[off[i] + k] = (uint8_t)s[k]; }
			if (qoff[i] >= 0) memcpy(&qual[qoff[i]], seqs[i].qual, seqs[i].l_seq);
			memcpy(&names[noff[i]], seqs[i].name, noff[i + 1] - noff[i]);
			if (any_comment && coff[i + 1] > coff[i]) memcpy(&comments[coff[i]], seqs[i].comment, coff[i + 1] - coff[i]);
		}
	});
	int rc = bwahip_batch_upload(c, n, codes.data(), off.data());
	if (rc) return rc;
	if ((rc = dev_upload(c->d_qual, qual.data(), qual.size(), c->stream)) || (rc = dev_upload(c->d_qual_off, qoff.data(), (size_t)n * 8, c->stream)) ||
	    (rc = dev_upload(c->d_names, names.data(), names.size(), c->stream)) || (rc = dev_upload(c->d_name_off, noff.data(), (size_t)(n + 1) * 8, c->stream)) ||
	    (rc = dev_upload(c->d_comment_off, coff.data(), (size_t)(n + 1) * 8, c->stream))) return rc;
	if (any_comment) { if ((rc = dev_upload(c->d_comments, comments.data(), comments.size(), c->stream))) return rc; }
	else c->d_comments.release();
	HIP_TRY(hipStreamSynchronize(c->stream));
	return 0;
}

extern "C" int bwahip_process_seqs(bwahip_ctx *ctx, const bwahip_opt_t *opt, int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0)
{
	if (!ctx || !opt || n < 0 || (n && !seqs)) return BWAHIP_EINVAL;
	const bool pe = (opt->flag & BWAHIP_F_PE) != 0;
	if (pe && (n & 1)) return BWAHIP_EINVAL;
	if (pe || !ctx->knobs.gpu_final) return bwahip_process_seqs_host(ctx, opt, n_processed, n, seqs, pes0);
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(ctx->device));
	int rc = upload_batch_text(ctx, opt->n_threads, n, seqs);
	if (rc) return rc;
	if ((rc = run_pipeline(ctx, opt, false, false))) return rc;
	if ((rc = run_final_se(ctx, opt, n_processed, false))) return rc;
	// SAM text back in one piece, then one malloc()ed string per read as the reference's contract wants (bwamem.c:1054)
	std::vector<int64_t> soff(n + 1);
	std::vector<char> text((size_t)ctx->total_sam + 1);
	// (on the context's stream: it is a non-blocking stream, a plain hipMemcpy would not wait for the SAM kernel)
	HIP_TRY(hipMemcpyAsync(soff.data(), ctx->d_sam_off.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (ctx->total_sam) HIP_TRY(hipMemcpyAsync(text.data(), ctx->d_sam.p, (size_t)ctx->total_sam, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	std::atomic<int> oom(0);
	par_for_chunks(n, opt->n_threads, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			const size_t len = (size_t)(soff[i + 1] - soff[i]);
			char *p = (char*)malloc(len + 1);
			if (!p) { oom = 1; seqs[i].sam = nullptr; continue; }
			memcpy(p, text.data() + soff[i], len); p[len] = 0;
			seqs[i].sam = p;
		}
	});
	return oom ? BWAHIP_ENOMEM : 0;
}
