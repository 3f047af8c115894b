// Host runtime of libbwahip: context, HBM residency of the index, batch buffers, kernel sequencing
// and the C ABI of include/bwahip.h.  One context drives one GPU from one host thread; multi-GPU
// runs use one process (one context) per GPU and shard whole batches (DESIGN.md section "Multi-GPU").
#include "ctx_internal.h"
#include <math.h>
#include <algorithm>
#include <atomic>

namespace {
// exclusive scan int32 -> int64 (n+1 outputs) in three small launches: tile sums, scan of the tile sums, tile scans
constexpr int SCAN_T = 256, SCAN_PER = 16, SCAN_TILE = SCAN_T * SCAN_PER;

__device__ __forceinline__ long long block_excl_scan(long long v, long long *sh, long long *total)
{
	const int t = threadIdx.x;
	sh[t] = v;
	__syncthreads();
	for (int d = 1; d < SCAN_T; d <<= 1) {
		long long o = t >= d ? sh[t - d] : 0;
		__syncthreads();
		sh[t] += o;
		__syncthreads();
	}
	const long long incl = sh[t];
	*total = sh[SCAN_T - 1];
	__syncthreads();
	return incl - v;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const int *in, long long *sums, int n)
{
	__shared__ long long sh[SCAN_T];
	const int b = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
	long long s = 0;
	for (int i = 0; i < SCAN_PER; ++i) if (b + i < n) s += in[b + i];
	long long total;
	block_excl_scan(s, sh, &total);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_top(long long *sums, int n_tiles, int64_t *out_total)
{
	__shared__ long long sh[SCAN_T];
	long long run = 0;
	for (int base = 0; base < n_tiles; base += SCAN_T) {
		const int i = base + threadIdx.x;
		const long long v = i < n_tiles ? sums[i] : 0;
		long long total;
		const long long ex = block_excl_scan(v, sh, &total);
		if (i < n_tiles) sums[i] = run + ex;
		run += total;
	}
	if (threadIdx.x == 0) *out_total = run;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_tiles(const int *in, const long long *sums, int64_t *out, int n)
{
	__shared__ long long sh[SCAN_T];
	const int b = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER;
	int v[SCAN_PER];
	long long s = 0;
#pragma unroll
	for (int i = 0; i < SCAN_PER; ++i) { v[i] = b + i < n ? in[b + i] : 0; s += v[i]; }
	long long total;
	long long run = sums[blockIdx.x] + block_excl_scan(s, sh, &total);
#pragma unroll
	for (int i = 0; i < SCAN_PER; ++i) { if (b + i < n) out[b + i] = run; run += v[i]; }
}

} // namespace

int launch_scan(const int *in, int64_t *out, int n, DevBuf &tmp, hipStream_t st)
{
	const int tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	int rc = tmp.ensure((size_t)(tiles + 1) * 8);
	if (rc) return rc;
	hipLaunchKernelGGL(k_scan_sums, dim3(tiles), dim3(SCAN_T), 0, st, in, tmp.as<long long>(), n);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(SCAN_T), 0, st, tmp.as<long long>(), tiles, out + n);
	hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(SCAN_T), 0, st, in, tmp.as<long long>(), out, n);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

static const char *g_kernel_names[] = { "k_smem", "k_scan", "k_seeds", "k_chain", "k_scan2", "k_extend", "k_smem_heavy", "k_extend_spec", "k_smem3", "k_intv_sort", "k_seed_sw",
                                         "pe_rescue", "k_mark", "k_cigar", "k_sam_size", "k_sam_write" };   // [11..15]: finalisation stages (bwahip_batch_run_sam)

DevOpt make_dev_opt(const bwahip_opt_t *o)
{
	DevOpt d;
	memset(&d, 0, sizeof d);
	d.a = o->a; d.b = o->b; d.o_del = o->o_del; d.e_del = o->e_del; d.o_ins = o->o_ins; d.e_ins = o->e_ins;
	d.pen_clip5 = o->pen_clip5; d.pen_clip3 = o->pen_clip3; d.w = o->w; d.zdrop = o->zdrop;
	d.min_seed_len = o->min_seed_len; d.split_width = o->split_width; d.max_occ = o->max_occ;
	d.max_chain_gap = o->max_chain_gap; d.max_mem_intv = (int)o->max_mem_intv;
	d.split_len = (int)(o->min_seed_len * o->split_factor + .499);          // bwamem.c:141
	d.min_chain_weight = o->min_chain_weight; d.max_chain_extend = o->max_chain_extend;
	d.mask_level = o->mask_level; d.drop_ratio = o->drop_ratio; d.mask_level_redun = o->mask_level_redun;
	memcpy(d.mat, o->mat, 25);
	d.T = o->T; d.flag = o->flag; d.pen_unpaired = o->pen_unpaired; d.max_ins = o->max_ins; d.max_matesw = o->max_matesw;
	d.max_XA_hits = o->max_XA_hits; d.max_XA_hits_alt = o->max_XA_hits_alt; d.mapQ_coef_fac = o->mapQ_coef_fac;
	d.XA_drop_ratio = o->XA_drop_ratio; d.mapQ_coef_len = o->mapQ_coef_len;
	return d;
}

extern "C" {

const char *bwahip_version(void) { return "bwahip 0.1 (gfx950)"; }

void bwahip_opt_init(bwahip_opt_t *o)        // mem_opt_init, bwamem.c:74-110
{
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 4;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100; o->T = 30; o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 5;
	o->max_mem_intv = 20;
	o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
	o->max_chain_gap = 10000; o->max_ins = 10000;
	o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->XA_drop_ratio = 0.80f;
	o->split_factor = 1.5f;
	o->chunk_size = 30000000;               // the fork's value (bwamem.c:99)
	o->n_threads = 1;
	o->max_XA_hits = 5; o->max_XA_hits_alt = 200;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95f;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log(o->mapQ_coef_len);
	for (int i = 0, k = 0; i < 5; ++i)       // bwa_fill_scmat, bwa.c:249
		for (int j = 0; j < 5; ++j) o->mat[k++] = (i == 4 || j == 4) ? -1 : i == j ? o->a : -o->b;
}

void bwahip_opt_fill_scmat(bwahip_opt_t *o)   // bwa_fill_scmat, bwa.c:249: call after changing a / b
{
	if (!o) return;
	int k = 0;
	for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) o->mat[k++] = i == j ? o->a : -o->b; o->mat[k++] = -1; }
	for (int j = 0; j < 5; ++j) o->mat[k++] = -1;
}

} // extern "C"
int dev_upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st)
{
	int rc = b.ensure(bytes ? bytes : 16);
	if (rc) return rc;
	if (bytes) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
	return 0;
}
static inline int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st) { return dev_upload(b, src, bytes, st); }
extern "C" {

int ctx_setup(bwahip_ctx *c, const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac)
{
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
	HIP_TRY(hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
	for (auto &e : c->ev_slice) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&c->ev_sam_half, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
	for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
	int rc;
	c->knobs.from_env();
	c->intv_cap = c->knobs.intv_cap;
	{
		std::vector<double> lt(BWAHIP_LOGTAB_N);
		for (int i = 0; i < BWAHIP_LOGTAB_N; ++i) lt[i] = i ? log((double)i) : 0.;   // glibc's log: the values the reference computes on the host
		if ((rc = upload(c->d_logtab, lt.data(), lt.size() * 8, c->stream))) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	if (c->external_index) {             // adopt caller-owned device arrays (e.g. received over RCCL)
		c->d_bwt.p = bwt->bwt; c->d_sa.p = bwt->sa; c->d_pac.p = (void*)pac;
	} else if (c->index_resident) {      // bwahip_init_rccl: the context's own buffers were filled by the broadcast
	} else {
		if ((rc = upload(c->d_bwt, bwt->bwt, bwt->bwt_size * 4, c->stream))) return rc;
		if ((rc = upload(c->d_sa, bwt->sa, bwt->n_sa * 8, c->stream))) return rc;
		if ((rc = upload(c->d_pac, pac, (size_t)bns->l_pac / 4 + 1, c->stream))) return rc;
	}
	std::vector<DevAnn> anns(bns->n_seqs);
	for (int i = 0; i < bns->n_seqs; ++i) anns[i] = { bns->anns[i].offset, bns->anns[i].len, bns->anns[i].is_alt };
	if ((rc = upload(c->d_anns, anns.data(), anns.size() * sizeof(DevAnn), c->stream))) return rc;
	if ((rc = c->d_misc.ensure(BWAHIP_MISC_BYTES))) return rc;
	HIP_TRY(hipMemsetAsync(c->d_misc.p, 0, BWAHIP_MISC_BYTES, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	DevIndex &ix = c->ix;
	memset(&ix, 0, sizeof ix);
	ix.bwt = c->d_bwt.as<uint4>(); ix.sa = c->d_sa.as<uint64_t>(); ix.pac = c->d_pac.as<uint8_t>(); ix.anns = c->d_anns.as<DevAnn>();
	ix.primary = bwt->primary; memcpy(ix.L2, bwt->L2, sizeof ix.L2); ix.seq_len = bwt->seq_len; ix.n_sa = bwt->n_sa;
	ix.l_pac = bns->l_pac; ix.sa_intv = bwt->sa_intv; ix.n_seqs = bns->n_seqs;
	for (ix.sa_shift = 0; (1 << ix.sa_shift) < bwt->sa_intv; ++ix.sa_shift);
	if ((1 << ix.sa_shift) != bwt->sa_intv) return BWAHIP_EINVAL;
	if (bwt->bwt_size < ((bwt->seq_len + 127) / 128) * 16) return BWAHIP_EINVAL;   // every 128-base block must be present
	// the bases of every Occ block as bit planes (fmi_dev.h): in place in the context's own array; a caller-owned array (bwahip_init_device)
	// is left as it is and a copy re-laid; a clone reads its source's array
	if (!c->share_from && !c->bwt_is_planes) {
		uint32_t *words = (uint32_t*)c->d_bwt.p;
		if (c->external_index) {
			if ((rc = c->d_bwtp.ensure((size_t)bwt->bwt_size * 4))) return rc;
			HIP_TRY(hipMemcpyAsync(c->d_bwtp.p, bwt->bwt, (size_t)bwt->bwt_size * 4, hipMemcpyDeviceToDevice, c->stream));
			words = (uint32_t*)c->d_bwtp.p;
			ix.bwt = c->d_bwtp.as<uint4>();
		}
		if ((rc = launch_bwt_planes(words, bwt->bwt_size, c->stream))) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	// the SA table of the kernels: every row (or every sa_intv-th), filled in on the GPU from the files' every 32nd (k_seed.hip)
	int want = c->knobs.sa_intv < 1 ? 1 : c->knobs.sa_intv;
	while (want & (want - 1)) want &= want - 1;
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	while (want < ix.sa_intv && ((size_t)(ix.seq_len / want) + 1) * 8 > free_b / 4) want <<= 1;
	if (want < ix.sa_intv) {
		const size_t n_dense = (size_t)(ix.seq_len / want) + 1;
		if ((rc = c->d_sa_dense.ensure(n_dense * 8))) return rc;
		if ((rc = launch_sa_densify(ix, c->d_sa_dense.as<uint64_t>(), want, c->stream))) return rc;
		HIP_TRY(hipStreamSynchronize(c->stream));
		ix.sa = c->d_sa_dense.as<uint64_t>(); ix.n_sa = n_dense; ix.sa_intv = want;
		for (ix.sa_shift = 0; (1 << ix.sa_shift) < want; ++ix.sa_shift);
	}
	// the interval table of the BWT search (k_smem.hip): shared by the clones of the context that built it
	if (c->share_from) { ix.kmer = c->share_from->ix.kmer; ix.kmer_k = c->share_from->ix.kmer_k; }
	else {
		int K = c->knobs.kmer_k > 16 ? 16 : c->knobs.kmer_k;
		int k_len = 1;
		while (k_len < 16 && (1ull << 2 * k_len) < ix.seq_len) ++k_len;          // longer strings than log4 of the text mostly do not occur
		if (K > k_len) K = k_len;
		HIP_TRY(hipMemGetInfo(&free_b, &total_b));
		while (K >= 2 && (kmer_off(K + 1) + 4) * 16 > free_b / 4) --K;
		if (K >= 2) {
			if ((rc = c->d_kmer.ensure((kmer_off(K + 1) + 4) * 16))) return rc;
			if ((rc = launch_kmer_table(ix, c->d_kmer.as<uint4>(), K, c->stream))) return rc;
			HIP_TRY(hipStreamSynchronize(c->stream));
			ix.kmer = c->d_kmer.as<uint4>(); ix.kmer_k = K;
		}
	}
	if (c->knobs.verbose) fprintf(stderr, "[bwahip] index in HBM: SA every %d rows (%.2f GB), interval table of strings up to %d bases (%.2f GB)\n", ix.sa_intv, ix.n_sa * 8 / 1e9, ix.kmer_k, ix.kmer_k ? (kmer_off(ix.kmer_k + 1) + 4) * 16 / 1e9 : 0.);
	return final_setup(c);
}

int bwahip_init(const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac, int device, bwahip_ctx **out)
{
	if (!bwt || !bns || !pac || !out || !bwt->bwt || !bwt->sa) return BWAHIP_EINVAL;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
		fprintf(stderr, "[bwahip] no usable HIP device (requested %d of %d)\n", device, n_dev);
		return BWAHIP_ENODEV;
	}
	bwahip_ctx *c = new bwahip_ctx();
	c->device = device;
	// the context keeps its own copy of the contig table and the packed reference (finalisation reads them); the caller's
	// arrays are not referenced after this returns
	int rc = bwahip_copy_host_index(bwt, bns, pac, &c->host);
	if (!rc) rc = ctx_setup(c, bwt, bns, pac);
	if (rc) { bwahip_destroy(c); return rc; }
	*out = c;
	return 0;
}

static int init_device_impl(const bwahip_bwt_t *bwt_dev, const bwahip_bns_t *bns, const uint8_t *pac_dev, int device, bwahip_ctx **out, const bwahip_ctx *share_from);
int bwahip_init_device(const bwahip_bwt_t *bwt_dev, const bwahip_bns_t *bns, const uint8_t *pac_dev, int device, bwahip_ctx **out)
{
	return init_device_impl(bwt_dev, bns, pac_dev, device, out, nullptr);
}
static int init_device_impl(const bwahip_bwt_t *bwt_dev, const bwahip_bns_t *bns, const uint8_t *pac_dev, int device, bwahip_ctx **out, const bwahip_ctx *share_from)
{
	if (!bwt_dev || !bns || !pac_dev || !out || !bwt_dev->bwt || !bwt_dev->sa) return BWAHIP_EINVAL;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
		fprintf(stderr, "[bwahip] no usable HIP device (requested %d of %d)\n", device, n_dev);
		return BWAHIP_ENODEV;
	}
	bwahip_ctx *c = new bwahip_ctx();
	c->device = device; c->external_index = true; c->share_from = share_from;
	// host copy of the contig table, and of the packed reference read back from the adopted device array (l_pac/4+1 bytes),
	// so that a rank that received its index over RCCL can run bwahip_process_seqs like the rank that loaded it
	int rc = bwahip_copy_host_index(bwt_dev, bns, nullptr, &c->host);
	if (!rc) rc = ctx_setup(c, bwt_dev, bns, pac_dev);
	if (!rc && hipMemcpy(c->host.pac, pac_dev, (size_t)bns->l_pac / 4 + 1, hipMemcpyDeviceToHost) != hipSuccess) rc = BWAHIP_ENODEV;
	if (rc) { bwahip_destroy(c); return rc; }
	*out = c;
	return 0;
}

// A further context on the same GPU that shares src's index arrays in HBM (nothing is copied on the device): its own streams,
// batch buffers and host thread.  Two contexts taking batches in turn keep the GPU busy through each other's latency-bound
// kernels and serial tails (double buffering).  src must outlive the clone.
int bwahip_ctx_clone(bwahip_ctx *src, bwahip_ctx **out)
{
	if (!src || !out || !src->d_bwt.p || !src->d_sa.p || !src->d_pac.p) return BWAHIP_EINVAL;
	bwahip_bwt_t b = src->host.bwt;
	b.bwt = (uint32_t*)const_cast<uint4*>(src->ix.bwt);       // (bit planes already: share_from tells ctx_setup)
	b.sa = const_cast<uint64_t*>(src->ix.sa); b.sa_intv = src->ix.sa_intv; b.n_sa = src->ix.n_sa;   // the table src's kernels read (the dense one when src built it)
	int rc = init_device_impl(&b, &src->host.bns, (const uint8_t*)src->d_pac.p, src->device, out, src);
	if (rc) return rc;
	(*out)->knobs = src->knobs; (*out)->intv_cap = src->knobs.intv_cap; (*out)->rg_id = src->rg_id;
	return 0;
}

int bwahip_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0; }

// A context on another GPU: the index arrays travel device to device (xGMI between the GPUs of a node), the new context owns its copy.
int bwahip_ctx_clone_on(bwahip_ctx *src, int device, bwahip_ctx **out)
{
	if (!src || !out || !src->d_bwt.p || !src->d_sa.p || !src->d_pac.p) return BWAHIP_EINVAL;
	if (device == src->device) return bwahip_ctx_clone(src, out);
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { fprintf(stderr, "[bwahip] no usable HIP device (requested %d of %d)\n", device, n_dev); return BWAHIP_ENODEV; }
	bwahip_ctx *c = new bwahip_ctx();
	c->device = device; c->index_resident = true; c->bwt_is_planes = true;
	const size_t n_bwt = (size_t)src->host.bwt.bwt_size * 4, n_sa = (size_t)src->host.bwt.n_sa * 8, n_pac = (size_t)src->host.bns.l_pac / 4 + 1;
	int rc = bwahip_copy_host_index(&src->host.bwt, &src->host.bns, src->host.pac, &c->host);
	if (!rc && hipSetDevice(device) != hipSuccess) rc = BWAHIP_ENODEV;
	if (!rc) rc = c->d_bwt.ensure(n_bwt);
	if (!rc) rc = c->d_sa.ensure(n_sa);
	if (!rc) rc = c->d_pac.ensure(n_pac);
	if (!rc) {
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src->device, 0);   // already enabled is fine
		(void)hipGetLastError();
		if (hipMemcpyPeer(c->d_bwt.p, device, src->ix.bwt, src->device, n_bwt) != hipSuccess || hipMemcpyPeer(c->d_sa.p, device, src->d_sa.p, src->device, n_sa) != hipSuccess ||
		    hipMemcpyPeer(c->d_pac.p, device, src->d_pac.p, src->device, n_pac) != hipSuccess) { fprintf(stderr, "[bwahip] device-to-device copy of the index failed: %s\n", hipGetErrorString(hipGetLastError())); rc = BWAHIP_ENODEV; }
	}
	if (!rc) rc = ctx_setup(c, &c->host.bwt, &c->host.bns, c->host.pac);
	if (rc) { bwahip_destroy(c); return rc; }
	c->knobs = src->knobs; c->intv_cap = src->knobs.intv_cap; c->rg_id = src->rg_id;
	*out = c;
	return 0;
}

int bwahip_init_from_files(const char *prefix, int device, bwahip_ctx **out)
{
	if (!prefix || !out) return BWAHIP_EINVAL;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
		fprintf(stderr, "[bwahip] no usable HIP device (requested %d of %d)\n", device, n_dev);
		return BWAHIP_ENODEV;
	}
	bwahip_ctx *c = new bwahip_ctx();
	c->device = device;
	int rc = bwahip_load_index_files(prefix, &c->host);
	if (!rc) rc = ctx_setup(c, &c->host.bwt, &c->host.bns, c->host.pac);
	if (rc) { bwahip_destroy(c); return rc; }
	*out = c;
	return 0;
}

void bwahip_destroy(bwahip_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	DevBuf *bufs[] = { &c->d_bwt, &c->d_bwtp, &c->d_sa, &c->d_sa_dense, &c->d_kmer, &c->d_pac, &c->d_anns, &c->d_seq, &c->d_off, &c->d_seq4, &c->d_smem_heavy, &c->d_raw, &c->d_raw_n, &c->d_intv, &c->d_intv_n, &c->d_seed_cnt,
	                   &c->d_lrep, &c->d_seed_base, &c->d_seeds, &c->d_scratch, &c->d_misc,
	                   &c->d_cw, &c->d_nxt, &c->d_ord, &c->d_wts, &c->d_kept, &c->d_first, &c->d_keep, &c->d_nodes, &c->d_stack,
	                   &c->d_chains, &c->d_chain_seeds, &c->d_chain_n, &c->d_kept_seeds, &c->d_reg_base, &c->d_regs, &c->d_tmp_regs,
	                   &c->d_reg_n, &c->d_srt, &c->d_dbg_chains, &c->d_dbg_seeds, &c->d_dbg_chain_n, &c->d_dbg_regs, &c->d_dbg_reg_n, &c->d_flt, &c->d_heavy, &c->d_perm, &c->d_spec_regs, &c->d_spec_items, &c->d_scan, &c->d_chain_big, &c->d_logtab, &c->d_redo, &c->d_big_t, &c->d_dedup, &c->d_cperm,
	                   &c->d_ctg_names, &c->d_ctg_name_off, &c->d_ctg_anno, &c->d_ctg_anno_off, &c->d_rg, &c->d_qual, &c->d_qual_off, &c->d_names, &c->d_name_off, &c->d_comments, &c->d_comment_off,
	                   &c->d_fregs, &c->d_fregs2, &c->d_fscr, &c->d_need, &c->d_xa_owner, &c->d_freg_n, &c->d_npri, &c->d_task_n, &c->d_rec_n, &c->d_task_base, &c->d_tasks, &c->d_aln_of_reg, &c->d_alns,
	                   &c->d_resc_flag, &c->d_zslab, &c->d_resc_ord, &c->d_pool, &c->d_fmisc, &c->d_fredo, &c->d_bigz, &c->d_rec_list, &c->d_xa_list, &c->d_sam_len, &c->d_sam_off, &c->d_sam,
	                   &c->d_hist, &c->d_pair_tab, &c->d_nb, &c->d_pe_cap, &c->d_pe_base, &c->d_pe_regs, &c->d_pe_n, &c->d_pe_tmp, &c->d_pe_keys, &c->d_pe_idx, &c->d_resc, &c->d_ms_slab, &c->d_pe_read, &c->d_sw_cnt, &c->d_sw_base, &c->d_sw_res, &c->d_sw_tasks, &c->d_sw_info, &c->d_task_lists };
	if (c->external_index) { c->d_bwt.p = nullptr; c->d_sa.p = nullptr; c->d_pac.p = nullptr; c->d_bwt.cap = c->d_sa.cap = c->d_pac.cap = 0; }
	for (DevBuf *b : bufs) b->release();
	c->h_stage.release(); c->h_sam.release(); c->h_sam2.release();
	for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	if (c->stream3) (void)hipStreamDestroy(c->stream3);
	if (c->stream_copy) (void)hipStreamDestroy(c->stream_copy);
	for (auto &e : c->ev_slice) if (e) (void)hipEventDestroy(e);
	if (c->ev_sam_half) (void)hipEventDestroy(c->ev_sam_half);
	if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if (c->ev_join) (void)hipEventDestroy(c->ev_join);
	bwahip_free_host_index(&c->host);
	delete c;
}

int bwahip_ctx_set_rg_id(bwahip_ctx *c, const char *id) { if (!c) return BWAHIP_EINVAL; c->rg_id = id ? id : ""; return 0; }
const char *bwahip_ctx_rg_id(const bwahip_ctx *c) { return c ? c->rg_id.c_str() : ""; }
const bwahip_bns_t *bwahip_bns(const bwahip_ctx *c) { return c ? &c->host.bns : nullptr; }
const bwahip_bwt_t *bwahip_bwt(const bwahip_ctx *c) { return c ? &c->host.bwt : nullptr; }
const uint8_t *bwahip_pac(const bwahip_ctx *c) { return c ? c->host.pac : nullptr; }

int bwahip_ctx_tune(bwahip_ctx *c, const char *key, int value)
{
	if (!c || !key) return BWAHIP_EINVAL;
	Knobs &k = c->knobs;
	if (!strcmp(key, "intv_cap")) { k.intv_cap = value < 2 ? 2 : value; c->intv_cap = k.intv_cap; }
	else if (!strcmp(key, "smem_lanes")) { if (value != 1 && value != 2 && value != 4 && value != 8) return BWAHIP_EINVAL; k.smem_lanes = value; }
	else if (!strcmp(key, "heavy_mult")) k.heavy_mult = value;
	else if (!strcmp(key, "chain_big_min")) k.chain_big_min = value;
	else if (!strcmp(key, "rank_sort_min")) k.rank_sort_min = value;
	else if (!strcmp(key, "spec_min_chains")) k.spec_min_chains = value;
	else if (!strcmp(key, "ext_lds_window")) k.ext_lds_window = value < 1 ? 1 : value;
	else if (!strcmp(key, "gpu_final")) k.gpu_final = value;
	else if (!strcmp(key, "gpu_pair")) k.gpu_pair = value;
	else if (!strcmp(key, "verbose")) k.verbose = value;
	else return BWAHIP_EINVAL;
	return 0;
}

// ------------------------------------------------------------------ known-answer helpers
int bwahip_kat_ksw_align(bwahip_ctx *c, int n, const int *params, const int8_t *mat25, const uint8_t *q, const int64_t *qoff,
                         const uint8_t *t, const int64_t *toff, int *out7)
{
	if (!c || n < 0 || (n && (!params || !q || !qoff || !t || !toff || !out7))) return BWAHIP_EINVAL;
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(c->device));
	bwahip_opt_t o; bwahip_opt_init(&o);
	if (mat25) memcpy(o.mat, mat25, 25);
	DevOpt dopt = make_dev_opt(&o);
	std::vector<int> items[2];                               // [0] word kernel, [1] byte kernel (KSW_XBYTE)
	size_t stride = 0;
	for (int i = 0; i < n; ++i) {
		const int qlen = params[8*i], tlen = params[8*i+1];
		if (qlen < 0 || tlen < 0 || qlen > (1 << 20) || tlen > (1 << 20)) return BWAHIP_EINVAL;
		items[(params[8*i+2] & 0x10000) ? 1 : 0].push_back(i);
		const size_t cells = (size_t)(qlen + 15) / 16 * 16;
		stride = std::max(stride, 5 * cells + 16 + 8 * cells + 2 * (size_t)tlen + 16);
	}
	stride = (stride + 15) & ~(size_t)15;
	DevBuf dp, dq, dqo, dt, dto, dout, dit, dw; int rc;
	if ((rc = upload(dp, params, (size_t)n * 32, c->stream)) || (rc = upload(dq, q, (size_t)qoff[n], c->stream)) || (rc = upload(dqo, qoff, (size_t)(n + 1) * 8, c->stream)) ||
	    (rc = upload(dt, t, (size_t)toff[n], c->stream)) || (rc = upload(dto, toff, (size_t)(n + 1) * 8, c->stream)) || (rc = dout.ensure((size_t)n * 28)) ||
	    (rc = dw.ensure(stride * (size_t)n))) goto done;
	for (int m = 0; m < 2 && !rc; ++m) {
		if (items[m].empty()) continue;
		if ((rc = upload(dit, items[m].data(), items[m].size() * 4, c->stream))) break;
		rc = launch_kat_align(dopt, (int)items[m].size(), m, dit.as<int>(), dp.as<int>(), dq.as<uint8_t>(), dqo.as<int64_t>(), dt.as<uint8_t>(), dto.as<int64_t>(),
		                      dw.as<uint8_t>(), stride, dout.as<int>(), c->stream);
		if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	}
	if (!rc && hipMemcpy(out7, dout.p, (size_t)n * 28, hipMemcpyDeviceToHost) != hipSuccess) rc = BWAHIP_ENODEV;
done:
	dp.release(); dq.release(); dqo.release(); dt.release(); dto.release(); dout.release(); dit.release(); dw.release();
	return rc;
}

// The region-list sorts (ks_introsort on mem_ars2 / mem_ars keys, bwamem.c:398-402) as the kernels run them: the wavefront's exact form and
// the one-lane restatement on the same n keys {k64, score, qb} (mode 0: by k64; mode 1: score desc, k64, qb).  idx_par / idx_seq: the two
// permutations (n ints each); status2[0] = 1 when the parallel form ran to the end (0: it fell back, idx_par is the identity), status2[1] != 0: error.
int bwahip_kat_introsort(bwahip_ctx *c, int n, int mode, const int64_t *k64, const int *score, const int *qb, int *idx_par, int *idx_seq, int *status2)
{
	if (!c || n < 1 || n > 65535 || (mode != 0 && mode != 1) || !k64 || !score || !qb || !idx_par || !idx_seq || !status2) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	struct K16 { int64_t k64; int score, qb; };
	std::vector<K16> keys(n);
	for (int i = 0; i < n; ++i) keys[i] = { k64[i], score[i], qb[i] };
	DevBuf dk, dp, ds, dw, dst; int rc;
	if ((rc = upload(dk, keys.data(), (size_t)n * 16, c->stream)) || (rc = dp.ensure((size_t)n * 8)) || (rc = ds.ensure((size_t)n * 8)) ||
	    (rc = dw.ensure(kat_isort_work_ints(n) * 4 + 64)) || (rc = dst.ensure(16))) goto done;
	rc = launch_kat_isort(n, mode, dk.p, dp.as<int>(), ds.as<int>(), dw.as<int>(), dst.as<int>(), c->stream);
	if (!rc && (hipMemcpyAsync(idx_par, dp.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipMemcpyAsync(idx_seq, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
	            hipMemcpyAsync(status2, dst.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess)) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
done:
	dk.release(); dp.release(); ds.release(); dw.release(); dst.release();
	return rc;
}

int bwahip_kat_occ4(bwahip_ctx *c, int n, const uint64_t *k, uint64_t *out)
{
	if (!c || n < 0) return BWAHIP_EINVAL;
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(c->device));
	DevBuf dk, dout; int rc;
	if ((rc = upload(dk, k, (size_t)n * 8, c->stream)) || (rc = dout.ensure((size_t)n * 32))) { dk.release(); dout.release(); return rc; }
	rc = launch_kat_occ4(c->ix, n, dk.as<uint64_t>(), dout.as<uint64_t>(), c->stream);
	if (!rc && hipMemcpyAsync(out, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	dk.release(); dout.release();
	return rc;
}

int bwahip_kat_sa(bwahip_ctx *c, int n, const uint64_t *k, uint64_t *out)
{
	if (!c || n < 0) return BWAHIP_EINVAL;
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(c->device));
	DevBuf dk, dout; int rc;
	if ((rc = upload(dk, k, (size_t)n * 8, c->stream)) || (rc = dout.ensure((size_t)n * 8))) { dk.release(); dout.release(); return rc; }
	rc = launch_kat_sa(c->ix, n, dk.as<uint64_t>(), dout.as<uint64_t>(), c->stream);
	if (!rc && hipMemcpyAsync(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	dk.release(); dout.release();
	return rc;
}

int bwahip_index_footprint(bwahip_ctx *c, int *sa_intv, int *kmer_k, uint64_t *bytes4)
{
	if (!c) return BWAHIP_EINVAL;
	if (sa_intv) *sa_intv = c->ix.sa_intv;
	if (kmer_k) *kmer_k = c->ix.kmer_k;
	if (bytes4) {
		bytes4[0] = (uint64_t)c->host.bwt.bwt_size * 4;                       // Occ-interleaved BWT
		bytes4[1] = (uint64_t)c->ix.n_sa * 8;                                  // the SA table the kernels read
		bytes4[2] = (uint64_t)c->host.bns.l_pac / 4 + 1;                       // packed reference
		bytes4[3] = c->ix.kmer_k ? (kmer_off(c->ix.kmer_k + 1) + 4) * 16 : 0;  // interval table
	}
	return 0;
}

int bwahip_kat_kmer_table(bwahip_ctx *c, int *k_out, uint64_t *bad_out)
{
	if (!c || !k_out || !bad_out) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	*k_out = c->ix.kmer_k; *bad_out = 0;
	if (c->ix.kmer_k < 2) return 0;
	DevBuf d; int rc;
	if ((rc = d.ensure(16))) return rc;
	HIP_TRY(hipMemsetAsync(d.p, 0, 16, c->stream));
	for (int L = 2; L <= c->ix.kmer_k && !rc; ++L) rc = launch_kmer_check(c->ix, L, d.as<unsigned long long>(), c->stream);
	unsigned long long bad = 0;
	if (!rc && hipMemcpyAsync(&bad, d.p, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	d.release();
	*bad_out = bad;
	return rc;
}

int bwahip_kat_extend(bwahip_ctx *c, int n, const uint64_t *ik3, const int *is_back, uint64_t *ok12)
{
	if (!c || n < 0) return BWAHIP_EINVAL;
	if (n == 0) return 0;
	HIP_TRY(hipSetDevice(c->device));
	DevBuf dk, db, dout; int rc;
	if ((rc = upload(dk, ik3, (size_t)n * 24, c->stream)) || (rc = upload(db, is_back, (size_t)n * 4, c->stream)) || (rc = dout.ensure((size_t)n * 96))) {
		dk.release(); db.release(); dout.release(); return rc;
	}
	rc = launch_kat_extend(c->ix, n, dk.as<uint64_t>(), db.as<int>(), dout.as<uint64_t>(), c->stream);
	if (!rc && hipMemcpyAsync(ok12, dout.p, (size_t)n * 96, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	dk.release(); db.release(); dout.release();
	return rc;
}

// ------------------------------------------------------------------ device-resident batch
int bwahip_batch_upload(bwahip_ctx *c, int n, const uint8_t *seq, const int64_t *off)
{
	if (!c || n < 0 || (n && (!seq || !off))) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	int max_len = 0;
	for (int i = 0; i < n; ++i) {
		int64_t l = off[i + 1] - off[i];
		if (l < 0) return BWAHIP_EINVAL;
		if (l > BWAHIP_MAX_READ_LEN) { fprintf(stderr, "[bwahip] read %d is %lld bases long (limit %d)\n", i, (long long)l, BWAHIP_MAX_READ_LEN); return BWAHIP_ECAPACITY; }
		if (l > max_len) max_len = (int)l;
	}
	c->n_reads = n; c->max_len = max_len; c->total_bases = n ? off[n] - off[0] : 0;
	int rc;
	if (n && off[0] != 0) return BWAHIP_EINVAL;
	if ((rc = upload(c->d_seq, seq, (size_t)c->total_bases, c->stream))) return rc;
	if ((rc = upload(c->d_off, off, (size_t)(n + 1) * 8, c->stream))) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return 0;
}

int bwahip_batch_attach(bwahip_ctx *c, int n, const uint8_t *seq_dev, const int64_t *off_dev, int max_len, int64_t total_bases)
{
	if (!c || n < 0 || (n && (!seq_dev || !off_dev)) || max_len < 0 || total_bases < 0) return BWAHIP_EINVAL;
	if (max_len > BWAHIP_MAX_READ_LEN) return BWAHIP_ECAPACITY;
	HIP_TRY(hipSetDevice(c->device));
	c->n_reads = n; c->max_len = max_len; c->total_bases = total_bases;
	c->d_seq.adopt((void*)seq_dev, (size_t)total_bases);
	c->d_off.adopt((void*)off_dev, (size_t)(n + 1) * 8);
	return 0;
}

int bwahip_n_kernels(void) { return (int)(sizeof(g_kernel_names) / sizeof(g_kernel_names[0])); }
const char *bwahip_kernel_name(int i) { return i >= 0 && i < bwahip_n_kernels() ? g_kernel_names[i] : ""; }

} // extern "C" (the pipeline itself has C++ linkage: final_rt.hip calls it)

int run_pipeline(bwahip_ctx *c, const bwahip_opt_t *opt, bool timed, bool dump)
{
	const bool verbose = c->knobs.verbose != 0;
#define STAGE_LOG(name) do { if (verbose) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[bwahip] %s done (%s)\n", name, hipGetErrorString(hipGetLastError())); fflush(stderr); } } while (0)
	const int n = c->n_reads;
	if (n == 0) return 0;
	DevOpt dopt = make_dev_opt(opt);
	if (c->ix.seq_len >= (1ull << 38) || c->max_len >= (1 << 14)) return BWAHIP_EINVAL;   // k_smem packs list entries as 3 x 38 + 14 bits
	unsigned long long *counters = c->d_misc.as<unsigned long long>();
	unsigned int *queue = (unsigned int*)(counters + (size_t)CNT_SLOTS * CNT_N);
	int *err = (int*)(queue + 4);
	int rc;
	for (int attempt = 0; attempt < 8; ++attempt) {
		const int cap = c->intv_cap, lcap = c->max_len + 2;
		const int G = c->knobs.smem_lanes;                       // lanes per read in k_smem (1, 2, 4 or 8)
		const int groups = smem_default_groups(G);
		if ((rc = c->d_intv.ensure((size_t)n * cap * sizeof(DevIntv))) || (rc = c->d_raw.ensure((size_t)n * cap * sizeof(DevIntv))) || (rc = c->d_raw_n.ensure((size_t)n * 4))) return rc;
		if ((rc = c->d_intv_n.ensure((size_t)n * 4)) || (rc = c->d_seed_cnt.ensure((size_t)n * 4)) || (rc = c->d_lrep.ensure((size_t)n * 4))) return rc;
		if ((rc = c->d_seed_base.ensure((size_t)(n + 1) * 8))) return rc;
		if ((rc = c->d_scratch.ensure((size_t)groups * (size_t)lcap * 16))) return rc;
		HIP_TRY(hipMemsetAsync(c->d_misc.p, 0, BWAHIP_MISC_BYTES, c->stream));
		SmemLaunch sl;
		memset(&sl, 0, sizeof sl);
		sl.ix = c->ix; sl.opt = dopt; sl.n_reads = n; sl.seq = c->d_seq.as<uint8_t>(); sl.off = c->d_off.as<int64_t>();
		sl.out = c->d_intv.as<DevIntv>(); sl.out_n = c->d_intv_n.as<int>(); sl.cap = cap;
		sl.raw = c->d_raw.as<DevIntv>(); sl.raw_n = c->d_raw_n.as<int>();
		sl.seed_cnt = c->d_seed_cnt.as<int>(); sl.l_rep = c->d_lrep.as<int>();
		sl.seq4_stride = (c->max_len + 15) / 16 + 1;            // +1: a word of 0xF past the longest read
		if ((rc = c->d_seq4.ensure((size_t)n * sl.seq4_stride * 8))) return rc;
		sl.seq4 = c->d_seq4.as<uint64_t>();
		if (attempt == 0 && (rc = launch_pack4(sl, c->stream))) return rc;
		sl.scratch = c->d_scratch.as<DevIntv>(); sl.lcap = lcap; sl.queue = queue; sl.counters = counters; sl.err = err; sl.groups_total = groups;
		const int heavy_mult = c->knobs.heavy_mult >= 0 ? c->knobs.heavy_mult : c->max_len <= 200 ? 10 : 30;   // x read length; 0 = never hand off
		if ((rc = c->d_smem_heavy.ensure((size_t)n * 4))) return rc;
		sl.heavy_list = c->d_smem_heavy.as<int>(); sl.heavy_n = queue + 1; sl.heavy_mult = heavy_mult; sl.worst_n = (int*)(queue + 2);
		if (timed) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
		if ((rc = launch_smem(sl, G, c->stream))) return rc;
		STAGE_LOG("k_smem");
		if (timed) HIP_TRY(hipEventRecord(c->ev[1], c->stream));
		if (heavy_mult > 0 && (rc = launch_smem_heavy(sl, c->stream))) return rc;
		STAGE_LOG("k_smem_heavy");
		if (timed) HIP_TRY(hipEventRecord(c->ev[12], c->stream));
		if ((rc = launch_smem3(sl, c->stream))) return rc;
		STAGE_LOG("k_smem3");
		if (timed) HIP_TRY(hipEventRecord(c->ev[13], c->stream));
		if ((rc = launch_intv_sort(sl, c->stream))) return rc;
		STAGE_LOG("k_intv_sort");
		if (timed) HIP_TRY(hipEventRecord(c->ev[10], c->stream));
		if ((rc = launch_scan(c->d_seed_cnt.as<int>(), c->d_seed_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
		if (timed) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
		// the number of seeds sizes the next buffers: one 8-byte read-back per batch
		int64_t total = 0; int h_err = 0, h_worst = 0;
		HIP_TRY(hipMemcpyAsync(&total, c->d_seed_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpyAsync(&h_err, err, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpyAsync(&h_worst, queue + 2, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (h_err) { fprintf(stderr, "[bwahip] k_smem reported an internal inconsistency\n"); return BWAHIP_EINTERNAL; }
		if (const char *dump_ext = c->knobs.dump_ext) {   // diagnostic: per-read bwt_extend counts as int32
			std::vector<int> h_e(n);
			HIP_TRY(hipMemcpy(h_e.data(), c->d_lrep.p, (size_t)n * 4, hipMemcpyDeviceToHost));
			if (FILE *fp = fopen(dump_ext, "wb")) { fwrite(h_e.data(), 4, n, fp); fclose(fp); }
		}
		// interval-list overflow: the kernel reports the largest true count; re-run the batch with more room (GPU only, no CPU path)
		if (h_worst > cap) { c->intv_cap = h_worst + 16; continue; }
		c->total_seeds = total;
		if ((rc = c->d_seeds.ensure((size_t)(total ? total : 1) * sizeof(DevSeed)))) return rc;
		SeedLaunch se;
		memset(&se, 0, sizeof se);
		se.ix = c->ix; se.opt = dopt; se.n_reads = n; se.off = c->d_off.as<int64_t>();
		se.intv = c->d_intv.as<DevIntv>(); se.intv_n = c->d_intv_n.as<int>(); se.cap = cap;
		se.seed_base = c->d_seed_base.as<int64_t>(); se.seeds = c->d_seeds.as<DevSeed>(); se.counters = counters;
		if (timed) HIP_TRY(hipEventRecord(c->ev[3], c->stream));
		if ((rc = launch_seeds(se, total, c->stream))) return rc;
		STAGE_LOG("k_seeds");
		if (timed) HIP_TRY(hipEventRecord(c->ev[4], c->stream));
		// ---- K3: chaining + chain filter
		const size_t T = (size_t)(total ? total : 1);
		if ((rc = c->d_cw.ensure(T * sizeof(ChainWOpaque))) || (rc = c->d_nxt.ensure(T * 4)) || (rc = c->d_ord.ensure(T * 4)) ||
		    (rc = c->d_wts.ensure(T * 4)) || (rc = c->d_kept.ensure(T * 4)) || (rc = c->d_first.ensure(T * 4)) || (rc = c->d_keep.ensure(T * 4)) ||
		    (rc = c->d_nodes.ensure(((T >> 2) + 4 * (size_t)n + 8) * sizeof(BtNodeOpaque))) || (rc = c->d_stack.ensure((size_t)n * 256 * 4)) ||
		    (rc = c->d_chains.ensure(T * sizeof(DevChain))) || (rc = c->d_chain_seeds.ensure(T * sizeof(DevSeed))) ||
		    (rc = c->d_chain_n.ensure((size_t)n * 4)) || (rc = c->d_kept_seeds.ensure((size_t)n * 4)) || (rc = c->d_reg_base.ensure((size_t)(n + 1) * 8)))
			return rc;
		if (dump && ((rc = c->d_dbg_chains.ensure(T * sizeof(DevChain))) || (rc = c->d_dbg_seeds.ensure(T * sizeof(DevSeed))) || (rc = c->d_dbg_chain_n.ensure((size_t)n * 4))))
			return rc;
		ChainLaunch cl;
		memset(&cl, 0, sizeof cl);
		cl.ix = c->ix; cl.opt = dopt; cl.n_reads = n; cl.off = c->d_off.as<int64_t>();
		cl.intv = c->d_intv.as<DevIntv>(); cl.intv_n = c->d_intv_n.as<int>(); cl.cap = cap;
		cl.seed_base = c->d_seed_base.as<int64_t>(); cl.seeds = c->d_seeds.as<DevSeed>();
		cl.cw_ = c->d_cw.as<ChainWOpaque>(); cl.nxt = c->d_nxt.as<int>(); cl.ord = c->d_ord.as<int>(); cl.wts = c->d_wts.as<int>();
		cl.kept = c->d_kept.as<int>(); cl.first = c->d_first.as<int>(); cl.keep_list = c->d_keep.as<int>();
		cl.nodes_ = c->d_nodes.as<BtNodeOpaque>(); cl.stack = c->d_stack.as<int>();
		cl.chains = c->d_chains.as<DevChain>(); cl.chain_seeds = c->d_chain_seeds.as<DevSeed>();
		cl.chain_n = c->d_chain_n.as<int>(); cl.kept_seeds = c->d_kept_seeds.as<int>();
		cl.counters = counters;
		if ((rc = c->d_flt.ensure(T * 32)) || (rc = c->d_heavy.ensure((size_t)(n + 4) * 4))) return rc;
		cl.flt = c->d_flt.as<int>(); cl.heavy_list = c->d_heavy.as<int>() + 4; cl.heavy_count = c->d_heavy.as<int>();
		const int big_min = c->knobs.chain_big_min;              // seeds; < 0 = off
		if (big_min >= 0) {
			if ((rc = c->d_chain_big.ensure(((size_t)4 * n + 8) * 4))) return rc;
			HIP_TRY(hipMemsetAsync(c->d_chain_big.p, 0, 32, c->stream));
			cl.big_list = c->d_chain_big.as<int>() + 8; cl.big_count = c->d_chain_big.as<int>(); cl.big_min = big_min; cl.big_max = std::min(c->knobs.chain_big_max, 3200); cl.mid_max = std::min(c->knobs.chain_mid_max, 1536); cl.glb_grid = std::max(c->knobs.chain_glb_grid, 64);   // 800 LDS nodes >= 0.24 x seeds (every node but the root holds >= 5 keys)
		}
		HIP_TRY(hipMemsetAsync(c->d_heavy.p, 0, 16, c->stream));
		if (dump) { cl.dbg_chains = c->d_dbg_chains.as<DevChain>(); cl.dbg_seeds = c->d_dbg_seeds.as<DevSeed>(); cl.dbg_chain_n = c->d_dbg_chain_n.as<int>(); }
		if (timed) HIP_TRY(hipEventRecord(c->ev[5], c->stream));
		if ((rc = launch_chain(cl, c->stream, c->stream2, c->stream3, c->ev_fork, c->ev_join, c->ev_join3))) return rc;
		if ((rc = launch_chain_flt(cl, c->stream))) return rc;
		STAGE_LOG("k_chain");
		if (timed) HIP_TRY(hipEventRecord(c->ev[6], c->stream));
		// ---- K3b: mem_flt_chained_seeds (bwamem.c:605).  With -W 0 it returns at its first test for every read of 2..700
		// bases (5.5 ln l > 0.05 l there), so the launch is skipped; the kernel itself repeats the test per read.
		if (opt->min_chain_weight != 0 || c->max_len > 700) {
			SeedSwLaunch ss;
			memset(&ss, 0, sizeof ss);
			ss.ix = c->ix; ss.opt = dopt; ss.n_reads = n; ss.seq = c->d_seq.as<uint8_t>(); ss.off = c->d_off.as<int64_t>();
			ss.seed_base = c->d_seed_base.as<int64_t>(); ss.chains = c->d_chains.as<DevChain>(); ss.chain_seeds = c->d_chain_seeds.as<DevSeed>();
			ss.chain_n = c->d_chain_n.as<int>(); ss.kept_seeds = c->d_kept_seeds.as<int>(); ss.logtab = c->d_logtab.as<double>();
			if ((rc = launch_seed_sw(ss, c->stream))) return rc;
			STAGE_LOG("k_seed_sw");
		}
		if (timed) HIP_TRY(hipEventRecord(c->ev[14], c->stream));
		if ((rc = launch_scan(c->d_kept_seeds.as<int>(), c->d_reg_base.as<int64_t>(), n, c->d_scan, c->stream))) return rc;
		if (timed) HIP_TRY(hipEventRecord(c->ev[7], c->stream));
		int64_t total_regs = 0;
		HIP_TRY(hipMemcpyAsync(&total_regs, c->d_reg_base.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->total_regs = total_regs;
		// ---- K4/K5: extension + sort/dedup/patch
		const size_t R = (size_t)(total_regs ? total_regs : 1);
		if ((rc = c->d_regs.ensure(R * sizeof(DevReg))) || (rc = c->d_tmp_regs.ensure(R * sizeof(DevReg))) || (rc = c->d_reg_n.ensure((size_t)n * 4)) ||
		    (rc = c->d_srt.ensure(T * 8))) return rc;
		if (dump && ((rc = c->d_dbg_regs.ensure(R * sizeof(DevReg))) || (rc = c->d_dbg_reg_n.ensure((size_t)n * 4)))) return rc;
		ExtLaunch el;
		memset(&el, 0, sizeof el);
		el.ix = c->ix; el.opt = dopt; el.n_reads = n; el.seq = c->d_seq.as<uint8_t>(); el.off = c->d_off.as<int64_t>();
		el.seed_base = c->d_seed_base.as<int64_t>(); el.chains = c->d_chains.as<DevChain>(); el.chain_seeds = c->d_chain_seeds.as<DevSeed>();
		el.chain_n = c->d_chain_n.as<int>(); el.reg_base = c->d_reg_base.as<int64_t>();
		el.regs = c->d_regs.as<DevReg>(); el.reg_n = c->d_reg_n.as<int>(); el.tmp_regs = c->d_tmp_regs.as<DevReg>(); el.srt = c->d_srt.as<int>();
		if (dump) { el.dbg_regs = c->d_dbg_regs.as<DevReg>(); el.dbg_reg_n = c->d_dbg_reg_n.as<int>(); }
		if ((rc = c->d_perm.ensure((size_t)(n + 8) * 4))) return rc;
		el.kept_seeds = c->d_kept_seeds.as<int>(); el.perm = c->d_perm.as<int>() + 8; el.perm_counts = c->d_perm.as<int>();
		el.counters = counters; el.err = err;
		if ((rc = c->d_redo.ensure((size_t)(n + 4) * 4)) || (rc = c->d_big_t.ensure((size_t)BWAHIP_EXT_BIG_GRID * (BWAHIP_EXT_BIG_T + 64)))) return rc;
		HIP_TRY(hipMemsetAsync(c->d_redo.p, 0, 16, c->stream));
		if ((rc = c->d_dedup.ensure(((size_t)3 * n + 4) * 4))) return rc;
		HIP_TRY(hipMemsetAsync(c->d_dedup.p, 0, 16, c->stream));
		el.dedup_n = c->d_dedup.as<int>(); el.dedup_list = c->d_dedup.as<int>() + 4;
		el.redo_n = c->d_redo.as<int>(); el.redo_list = c->d_redo.as<int>() + 4; el.big_t = c->d_big_t.as<uint8_t>(); el.lds_window = c->knobs.ext_lds_window;
		el.rank_sort_min = c->knobs.rank_sort_min;
		const int spec_min = c->knobs.spec_min_chains;           // 0 = no ahead-of-time extension
		if (spec_min > 0) {
			if ((rc = c->d_spec_regs.ensure(T * sizeof(DevReg))) || (rc = c->d_spec_items.ensure(T * 8 + 16))) return rc;
			el.spec_regs = c->d_spec_regs.as<DevReg>(); el.spec_n = c->d_spec_items.as<int>(); el.spec_items = (int2*)(c->d_spec_items.as<int>() + 4);
			el.spec_min_chains = spec_min;
		}
		if (timed) HIP_TRY(hipEventRecord(c->ev[11], c->stream));
		if ((rc = launch_extend_spec(el, c->max_len, c->stream))) return rc;
		STAGE_LOG("k_extend_spec");
		if (timed) HIP_TRY(hipEventRecord(c->ev[8], c->stream));
		if (verbose) fprintf(stderr, "[bwahip] seeds=%lld regs_cap=%lld\n", (long long)total, (long long)total_regs);
		if ((rc = launch_extend(el, c->max_len, c->stream, c->stream2, c->ev_fork, c->ev_join))) return rc;
		STAGE_LOG("k_extend");
		if (timed) HIP_TRY(hipEventRecord(c->ev[9], c->stream));
		int h_err2[2] = { 0, 0 };
		HIP_TRY(hipMemcpyAsync(h_err2, err, 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		h_err = h_err2[0];
		if (h_err == 3 || h_err == 4) {
			fprintf(stderr, "[bwahip] read %d of the batch: a chain's reference window exceeds %d bases (%s)\n", h_err2[1], BWAHIP_EXT_BIG_T, h_err == 3 ? "extension" : "region patch");
			return BWAHIP_ECAPACITY;
		}
		if (h_err) { fprintf(stderr, "[bwahip] extension kernel reported code %d\n", h_err); return BWAHIP_EINTERNAL; }
		if (timed) {
			HIP_TRY(hipEventElapsedTime(&c->last_ms[0], c->ev[0], c->ev[1]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[1], c->ev[10], c->ev[2]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[6], c->ev[1], c->ev[12]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[8], c->ev[12], c->ev[13]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[9], c->ev[13], c->ev[10]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[2], c->ev[3], c->ev[4]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[3], c->ev[5], c->ev[6]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[4], c->ev[14], c->ev[7]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[10], c->ev[6], c->ev[14]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[5], c->ev[8], c->ev[9]));
			HIP_TRY(hipEventElapsedTime(&c->last_ms[7], c->ev[11], c->ev[8]));
		}
		return 0;
	}
	return BWAHIP_EINTERNAL;
}

extern "C" {

int bwahip_batch_run(bwahip_ctx *c, const bwahip_opt_t *opt, float *kernel_ms, int n_ms)
{
	if (!c || !opt) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	int rc = run_pipeline(c, opt, true, false);
	if (!rc && kernel_ms) for (int i = 0; i < n_ms && i < bwahip_n_kernels(); ++i) kernel_ms[i] = c->last_ms[i];
	return rc;
}

// The text the SAM stage prints for the attached batch, already in HBM: qualities (read r at qual_dev + qual_off_dev[r], < 0: none),
// NUL-terminated names (read r at names_dev + name_off_dev[r]; the buffer must extend 64 bytes past the last name).
int bwahip_batch_attach_text(bwahip_ctx *c, const uint8_t *qual_dev, const int64_t *qual_off_dev, const uint8_t *names_dev, const int64_t *name_off_dev)
{
	if (!c || !names_dev || !name_off_dev || (qual_dev && !qual_off_dev)) return BWAHIP_EINVAL;
	const int n = c->n_reads;
	c->d_qual.adopt((void*)qual_dev, 0); c->d_qual_off.adopt((void*)qual_off_dev, (size_t)n * 8);
	c->d_names.adopt((void*)names_dev, 0); c->d_name_off.adopt((void*)name_off_dev, (size_t)(n + 1) * 8);
	c->d_comments.release();
	int rc = c->d_comment_off.ensure((size_t)(n + 1) * 8);
	if (rc) return rc;
	HIP_TRY(hipSetDevice(c->device));
	HIP_TRY(hipMemsetAsync(c->d_comment_off.p, 0, (size_t)(n + 1) * 8, c->stream));
	return 0;
}

// mem_process_seqs over the attached batch, everything on the GPU: hot path, then finalisation and SAM text, which stays in HBM
// (bwahip_batch_sam downloads it).  kernel_ms as bwahip_batch_run, entries 11..15 = the finalisation stages.
int bwahip_batch_run_sam(bwahip_ctx *c, const bwahip_opt_t *opt, int64_t n_processed, const bwahip_pestat_t *pes0, float *kernel_ms, int n_ms)
{
	if (!c || !opt) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	int rc = run_pipeline(c, opt, true, false);
	if (!rc) rc = run_final(c, opt, n_processed, pes0, true);
	if (!rc && kernel_ms) {
		for (int i = 0; i < n_ms && i < 11; ++i) kernel_ms[i] = c->last_ms[i];
		if (n_ms > 11) kernel_ms[11] = c->final_ms[4];
		for (int i = 0; i < 4 && 12 + i < n_ms; ++i) kernel_ms[12 + i] = c->final_ms[i];
	}
	return rc;
}

// SAM text of the last bwahip_batch_run_sam: *out = malloc()ed buffer of *out_len bytes (reads in order); off (may be NULL):
// n + 1 offsets of the reads' texts
int bwahip_batch_sam(bwahip_ctx *c, char **out, int64_t *out_len, int64_t *off)
{
	if (!c || !out || !out_len) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	char *buf = (char*)malloc((size_t)c->total_sam + 1);
	if (!buf) return BWAHIP_ENOMEM;
	if (c->total_sam) HIP_TRY(hipMemcpyAsync(buf, c->d_sam.p, (size_t)c->total_sam, hipMemcpyDeviceToHost, c->stream));
	if (off && c->n_reads) HIP_TRY(hipMemcpyAsync(off, c->d_sam_off.p, (size_t)(c->n_reads + 1) * 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	buf[c->total_sam] = 0;
	*out = buf; *out_len = c->total_sam;
	return 0;
}

int bwahip_batch_counters(bwahip_ctx *c, uint64_t *counters, int n)
{
	if (!c || !counters || n < 0) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	std::vector<uint64_t> rows((size_t)CNT_SLOTS * CNT_N);
	HIP_TRY(hipMemcpy(rows.data(), c->d_misc.p, rows.size() * 8, hipMemcpyDeviceToHost));
	uint64_t h[CNT_N] = { 0 };
	for (int s = 0; s < CNT_SLOTS; ++s)
		for (int i = 0; i < CNT_N; ++i) {
			const uint64_t v = rows[(size_t)s * CNT_N + i];
			const bool is_max = i == CNT_MAX_EXT || (i >= 8 && i <= 15) || (i >= 21 && i <= 23);   // the *_max entries
			h[i] = is_max ? std::max(h[i], v) : h[i] + v;
		}
	for (int i = 0; i < n; ++i) counters[i] = i < CNT_N ? h[i] : 0;
	return 0;
}

static int batch_download_mt(bwahip_ctx *c, bwahip_alnreg_v *out, int nt)
{
	if (!c || !out) return BWAHIP_EINVAL;
	HIP_TRY(hipSetDevice(c->device));
	const int n = c->n_reads;
	if (n == 0) return 0;
	std::vector<int> h_regn(n);
	std::vector<int64_t> h_rbase(n + 1);
	std::vector<DevReg> h_regs((size_t)c->total_regs + 1);
	HIP_TRY(hipMemcpy(h_regn.data(), c->d_reg_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(h_rbase.data(), c->d_reg_base.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
	if (c->total_regs) HIP_TRY(hipMemcpy(h_regs.data(), c->d_regs.p, (size_t)c->total_regs * sizeof(DevReg), hipMemcpyDeviceToHost));
	std::atomic<int> oom(0);
	par_for_chunks(n, nt, [&](int64_t b, int64_t e) {
		for (int64_t i = b; i < e; ++i) {
			const int m = h_regn[i];
			out[i].n = out[i].m = m;
			out[i].a = m ? (bwahip_alnreg_t*)calloc(m, sizeof(bwahip_alnreg_t)) : nullptr;   // kv_init'ed vector when empty (bwamem.c:1075)
			if (m && !out[i].a) { oom = 1; out[i].n = out[i].m = 0; continue; }
			for (int k = 0; k < m; ++k) {
				const DevReg &p = h_regs[h_rbase[i] + k];
				bwahip_alnreg_t &q = out[i].a[k];              // memset(0) + the fields mem_chain2aln / dedup set
				q.rb = p.rb; q.re = p.re; q.frac_rep = p.frac_rep; q.qb = p.qb; q.qe = p.qe; q.rid = p.rid; q.score = p.score; q.truesc = p.truesc;
				q.sub = p.sub; q.csub = p.csub; q.sub_n = p.sub_n; q.w = p.w; q.seedcov = p.seedcov; q.seedlen0 = p.seedlen0;
				q.n_comp = p.n_comp; q.is_alt = p.is_alt;
			}
		}
	});
	return oom ? BWAHIP_ENOMEM : 0;
}
int bwahip_batch_download(bwahip_ctx *c, bwahip_alnreg_v *out) { return batch_download_mt(c, out, 1); }

// ------------------------------------------------------------------ stage dump (i64 records)
static void rec(std::vector<int64_t> &o, int64_t tag, const std::vector<int64_t> &v)
{
	o.push_back(tag); o.push_back((int64_t)v.size());
	o.insert(o.end(), v.begin(), v.end());
}

int bwahip_run_stages(bwahip_ctx *c, const bwahip_opt_t *opt, int n, const uint8_t *seq, const int64_t *off,
                      int stage_mask, int64_t **out, int64_t *out_len)
{
	if (!c || !opt || !out || !out_len) return BWAHIP_EINVAL;
	int rc = bwahip_batch_upload(c, n, seq, off);
	if (rc) return rc;
	if ((rc = run_pipeline(c, opt, false, true))) return rc;
	std::vector<int64_t> o;
	std::vector<int> h_n(n ? n : 1);
	std::vector<DevIntv> h_iv((size_t)n * c->intv_cap + 1);
	if (n) {
		HIP_TRY(hipMemcpy(h_n.data(), c->d_intv_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_iv.data(), c->d_intv.p, (size_t)n * c->intv_cap * sizeof(DevIntv), hipMemcpyDeviceToHost));
	}
	std::vector<int64_t> h_base(n + 1, 0);
	std::vector<DevSeed> h_seeds((size_t)c->total_seeds + 1);
	if (n) {
		HIP_TRY(hipMemcpy(h_base.data(), c->d_seed_base.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
		if (c->total_seeds) HIP_TRY(hipMemcpy(h_seeds.data(), c->d_seeds.p, (size_t)c->total_seeds * sizeof(DevSeed), hipMemcpyDeviceToHost));
	}
	auto f2i = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (int64_t)u; };
	const size_t T = (size_t)c->total_seeds + 1, R = (size_t)c->total_regs + 1;
	std::vector<DevChain> h_ch(T), h_dch(T);
	std::vector<DevSeed> h_cs(T), h_dcs(T);
	std::vector<int> h_chn(n + 1), h_dchn(n + 1), h_regn(n + 1), h_dregn(n + 1);
	std::vector<int64_t> h_rbase(n + 1, 0);
	std::vector<DevReg> h_regs(R), h_dregs(R);
	if (n) {
		if (c->total_seeds) {
			HIP_TRY(hipMemcpy(h_ch.data(), c->d_chains.p, (size_t)c->total_seeds * sizeof(DevChain), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_cs.data(), c->d_chain_seeds.p, (size_t)c->total_seeds * sizeof(DevSeed), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_dch.data(), c->d_dbg_chains.p, (size_t)c->total_seeds * sizeof(DevChain), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_dcs.data(), c->d_dbg_seeds.p, (size_t)c->total_seeds * sizeof(DevSeed), hipMemcpyDeviceToHost));
		}
		HIP_TRY(hipMemcpy(h_chn.data(), c->d_chain_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_dchn.data(), c->d_dbg_chain_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_regn.data(), c->d_reg_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_dregn.data(), c->d_dbg_reg_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_rbase.data(), c->d_reg_base.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost));
		if (c->total_regs) {
			HIP_TRY(hipMemcpy(h_regs.data(), c->d_regs.p, (size_t)c->total_regs * sizeof(DevReg), hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_dregs.data(), c->d_dbg_regs.p, (size_t)c->total_regs * sizeof(DevReg), hipMemcpyDeviceToHost));
		}
	}
	auto put_chains = [&](std::vector<int64_t> &v, int cnt, const DevChain *ch, const DevSeed *sd) {
		v.push_back(cnt);
		for (int k = 0; k < cnt; ++k) {
			const DevChain &h = ch[k];
			v.push_back(h.pos); v.push_back(h.rid); v.push_back(h.is_alt); v.push_back(h.w); v.push_back(h.kept);
			v.push_back(h.first); v.push_back(f2i(h.frac_rep)); v.push_back(h.n);
			for (int t = 0; t < h.n; ++t) { const DevSeed &d = sd[h.seed_off + t]; v.push_back(d.rbeg); v.push_back(d.qbeg); v.push_back(d.len); v.push_back(d.score); }
		}
	};
	auto put_regs = [&](std::vector<int64_t> &v, int cnt, const DevReg *rg) {
		v.push_back(cnt);
		for (int k = 0; k < cnt; ++k) {
			const DevReg &p = rg[k];
			v.push_back(p.rb); v.push_back(p.re); v.push_back(p.qb); v.push_back(p.qe); v.push_back(p.rid);
			v.push_back(p.score); v.push_back(p.truesc); v.push_back(p.sub); v.push_back(0); v.push_back(p.csub);
			v.push_back(p.sub_n); v.push_back(p.w); v.push_back(p.seedcov); v.push_back(0);
			v.push_back(0); v.push_back(p.seedlen0); v.push_back(p.n_comp); v.push_back(p.is_alt);
			v.push_back(f2i(p.frac_rep));
		}
	};
	for (int i = 0; i < n; ++i) {
		std::vector<int64_t> v = { (int64_t)i, off[i + 1] - off[i] };
		rec(o, 100, v);
		if (stage_mask & (1 << BWAHIP_STAGE_INTV)) {
			v.clear();
			for (int t = 0; t < h_n[i]; ++t) {
				const DevIntv &p = h_iv[(size_t)i * c->intv_cap + t];
				v.push_back((int64_t)p.x0); v.push_back((int64_t)p.x1); v.push_back((int64_t)p.x2); v.push_back((int64_t)p.info);
			}
			rec(o, BWAHIP_STAGE_INTV, v);
		}
		if (stage_mask & (1 << 6)) {                 // debug stage: raw seeds (rbeg,qbeg,len,rid) in look-up order
			v.clear();
			for (int64_t t = h_base[i]; t < h_base[i + 1]; ++t) {
				v.push_back(h_seeds[t].rbeg); v.push_back(h_seeds[t].qbeg); v.push_back(h_seeds[t].len); v.push_back(h_seeds[t].rid);
			}
			rec(o, 6, v);
		}
		if (stage_mask & (1 << BWAHIP_STAGE_CHAIN)) { v.clear(); put_chains(v, h_dchn[i], &h_dch[h_base[i]], &h_dcs[h_base[i]]); rec(o, BWAHIP_STAGE_CHAIN, v); }
		if (stage_mask & (1 << BWAHIP_STAGE_CHAIN_FLT)) { v.clear(); put_chains(v, h_chn[i], &h_ch[h_base[i]], &h_cs[h_base[i]]); rec(o, BWAHIP_STAGE_CHAIN_FLT, v); }
		if (stage_mask & (1 << BWAHIP_STAGE_REGS_PRE)) { v.clear(); put_regs(v, h_dregn[i], &h_dregs[h_rbase[i]]); rec(o, BWAHIP_STAGE_REGS_PRE, v); }
		if (stage_mask & (1 << BWAHIP_STAGE_REGS)) { v.clear(); put_regs(v, h_regn[i], &h_regs[h_rbase[i]]); rec(o, BWAHIP_STAGE_REGS, v); }
	}
	*out_len = (int64_t)o.size();
	*out = (int64_t*)malloc(o.size() * 8 + 8);
	if (!*out) return BWAHIP_ENOMEM;
	memcpy(*out, o.data(), o.size() * 8);
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

int bwahip_align_batch(bwahip_ctx *c, const bwahip_opt_t *opt, int n, bwahip_seq_t *seqs, bwahip_alnreg_v *regs_out)
{
	if (!c || !opt || n < 0 || (n && (!seqs || !regs_out))) return BWAHIP_EINVAL;
	std::vector<int64_t> off(n + 1, 0);
	for (int i = 0; i < n; ++i) { if (seqs[i].l_seq < 0) return BWAHIP_EINVAL; off[i + 1] = off[i] + seqs[i].l_seq; }
	std::vector<uint8_t> codes((size_t)off[n] + 1);
	par_for_chunks(n, opt->n_threads, [&](int64_t b, int64_t e) {   // in-place conversion exactly as bwamem.c:1067-1068
		for (int64_t i = b; i < e; ++i) {
			char *s = seqs[i].seq;
			for (int k = 0; k < seqs[i].l_seq; ++k) { s[k] = s[k] < 4 ? s[k] : (char)k_nt4[(uint8_t)s[k]]; codes[off[i] + k] = (uint8_t)s[k]; }
		}
	});
	int rc = bwahip_batch_upload(c, n, codes.data(), off.data());
	if (rc) return rc;
	if ((rc = run_pipeline(c, opt, false, false))) return rc;
	return batch_download_mt(c, regs_out, opt->n_threads);
}
int bwahip_kat_ksw_extend(bwahip_ctx *c, int n, const int *params, const uint8_t *q, const int64_t *qoff, const uint8_t *t, const int64_t *toff, int *out6)
{
	if (!c || n < 0 || (n && (!params || !q || !qoff || !t || !toff || !out6))) return BWAHIP_EINVAL;
	if (n == 0) return 0;
	for (int i = 0; i < n; ++i)
		if (params[10*i] < 0 || params[10*i] > BWAHIP_MAX_READ_LEN || params[10*i+1] < 0 || params[10*i+1] > BWAHIP_MAX_READ_LEN + 1400 || params[10*i+3] <= 0) return BWAHIP_ECAPACITY;
	HIP_TRY(hipSetDevice(c->device));
	bwahip_opt_t o; bwahip_opt_init(&o);
	DevOpt dopt = make_dev_opt(&o);
	DevBuf dp, dq, dqo, dt, dto, dout; int rc;
	if ((rc = upload(dp, params, (size_t)n * 40, c->stream)) || (rc = upload(dq, q, (size_t)qoff[n], c->stream)) || (rc = upload(dqo, qoff, (size_t)(n + 1) * 8, c->stream)) ||
	    (rc = upload(dt, t, (size_t)toff[n], c->stream)) || (rc = upload(dto, toff, (size_t)(n + 1) * 8, c->stream)) || (rc = dout.ensure((size_t)n * 24))) goto done;
	rc = launch_kat_ksw(dopt, n, dp.as<int>(), dq.as<uint8_t>(), dqo.as<int64_t>(), dt.as<uint8_t>(), dto.as<int64_t>(), dout.as<int>(), c->stream);
	if (!rc && hipMemcpyAsync(out6, dout.p, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
	if (hipStreamSynchronize(c->stream) != hipSuccess) rc = BWAHIP_ENODEV;
done:
	dp.release(); dq.release(); dqo.release(); dt.release(); dto.release(); dout.release();
	return rc;
}

} // extern "C"
