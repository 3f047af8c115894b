// The system's one collective behind the C ABI: rank 0 loads the index files, every rank of the node receives the three
// index arrays (bwt, sa, pac) and the contig table over RCCL (xGMI between the GPUs of a node) straight into its own HBM
// and builds its context on them.  Replaces transferIndex() of the reference (cuda/streams.cu:8), which copies one index
// to one GPU from the host.  RCCL is opened with dlopen at first use: libbwahip.so has no link-time dependency on it, and a
// process that already carries an RCCL (PyTorch) keeps using that one.
#include "ctx_internal.h"
#include <dlfcn.h>

namespace {

typedef struct { char internal[128]; } UniqueId;              // ncclUniqueId (rccl.h:43)
typedef void *Comm;
struct Rccl {
	void *h = nullptr;
	int (*GetUniqueId)(UniqueId*) = nullptr;
	int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
	int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
	int (*CommDestroy)(Comm) = nullptr;
	int (*CommAbort)(Comm) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool load()
	{
		if (h) return true;
		for (const char *nm : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
		if (!h) { fprintf(stderr, "[bwahip] cannot open librccl: %s\n", dlerror()); return false; }
		GetUniqueId = (int (*)(UniqueId*))dlsym(h, "ncclGetUniqueId");
		CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
		Broadcast = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t))dlsym(h, "ncclBroadcast");
		CommDestroy = (int (*)(Comm))dlsym(h, "ncclCommDestroy");
		CommAbort = (int (*)(Comm))dlsym(h, "ncclCommAbort");
		GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
		return GetUniqueId && CommInitRank && Broadcast && CommDestroy;
	}
};
Rccl g_rccl;
constexpr int NCCL_UINT8 = 1;                                 // ncclUint8 (rccl.h ncclDataType_t)

#define RCCL_TRY(expr) do { int e_ = (expr); if (e_ != 0) { \
	fprintf(stderr, "[bwahip] %s failed: %s\n", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "?"); rc = BWAHIP_ENODEV; goto done; } } while (0)

// contig table + holes as one byte blob: header, per contig {offset,len,n_ambs,gi,is_alt,name_len,anno_len}, strings, holes
struct MetaHdr { uint64_t primary, L2[5], seq_len, bwt_size, n_sa; int64_t l_pac; int32_t sa_intv, n_seqs, n_holes; uint32_t seed; };
std::vector<uint8_t> pack_meta(const HostIndex &h)
{
	std::vector<uint8_t> b;
	auto put = [&](const void *p, size_t n) { const uint8_t *q = (const uint8_t*)p; b.insert(b.end(), q, q + n); };
	MetaHdr m;
	memset(&m, 0, sizeof m);
	m.primary = h.bwt.primary; memcpy(m.L2, h.bwt.L2, sizeof m.L2); m.seq_len = h.bwt.seq_len; m.bwt_size = h.bwt.bwt_size; m.n_sa = h.bwt.n_sa;
	m.l_pac = h.bns.l_pac; m.sa_intv = h.bwt.sa_intv; m.n_seqs = h.bns.n_seqs; m.n_holes = h.bns.n_holes; m.seed = h.bns.seed;
	put(&m, sizeof m);
	for (int i = 0; i < h.bns.n_seqs; ++i) {
		const bwahip_ann_t &a = h.bns.anns[i];
		const int32_t nl = (int32_t)strlen(a.name ? a.name : ""), al = (int32_t)strlen(a.anno ? a.anno : "");
		put(&a.offset, 8); put(&a.len, 4); put(&a.n_ambs, 4); put(&a.gi, 4); put(&a.is_alt, 4); put(&nl, 4); put(&al, 4);
		put(a.name ? a.name : "", nl); put(a.anno ? a.anno : "", al);
	}
	for (int i = 0; i < h.bns.n_holes; ++i) { put(&h.bns.ambs[i].offset, 8); put(&h.bns.ambs[i].len, 4); put(&h.bns.ambs[i].amb, 1); }
	return b;
}
bool unpack_meta(const std::vector<uint8_t> &b, HostIndex *h)
{
	size_t o = 0;
	auto get = [&](void *p, size_t n) { if (o + n > b.size()) return false; memcpy(p, &b[o], n); o += n; return true; };
	MetaHdr m;
	memset(h, 0, sizeof *h);
	h->owned = true;
	if (!get(&m, sizeof m)) return false;
	h->bwt.primary = m.primary; memcpy(h->bwt.L2, m.L2, sizeof m.L2); h->bwt.seq_len = m.seq_len; h->bwt.bwt_size = m.bwt_size; h->bwt.n_sa = m.n_sa; h->bwt.sa_intv = m.sa_intv;
	h->bns.l_pac = m.l_pac; h->bns.n_seqs = m.n_seqs; h->bns.n_holes = m.n_holes; h->bns.seed = m.seed;
	h->bns.anns = (bwahip_ann_t*)calloc(m.n_seqs > 0 ? m.n_seqs : 1, sizeof(bwahip_ann_t));
	for (int i = 0; i < m.n_seqs; ++i) {
		bwahip_ann_t &a = h->bns.anns[i];
		int32_t nl, al;
		if (!get(&a.offset, 8) || !get(&a.len, 4) || !get(&a.n_ambs, 4) || !get(&a.gi, 4) || !get(&a.is_alt, 4) || !get(&nl, 4) || !get(&al, 4)) return false;
		a.name = (char*)calloc(nl + 1, 1); a.anno = (char*)calloc(al + 1, 1);
		if (!get(a.name, nl) || !get(a.anno, al)) return false;
	}
	h->bns.ambs = m.n_holes ? (bwahip_amb_t*)calloc(m.n_holes, sizeof(bwahip_amb_t)) : nullptr;
	for (int i = 0; i < m.n_holes; ++i) if (!get(&h->bns.ambs[i].offset, 8) || !get(&h->bns.ambs[i].len, 4) || !get(&h->bns.ambs[i].amb, 1)) return false;
	return true;
}

} // namespace

extern "C" int bwahip_rccl_unique_id(void *id128)
{
	if (!id128 || !g_rccl.load()) return BWAHIP_ENODEV;
	UniqueId id;
	if (g_rccl.GetUniqueId(&id) != 0) return BWAHIP_ENODEV;
	memcpy(id128, &id, 128);
	return 0;
}

extern "C" int bwahip_init_rccl(const char *prefix, int rank, int world, const void *id128, int device, bwahip_ctx **out)
{
	if (!out || !id128 || rank < 0 || world < 1 || rank >= world || (rank == 0 && !prefix)) return BWAHIP_EINVAL;
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { fprintf(stderr, "[bwahip] no usable HIP device (requested %d of %d)\n", device, n_dev); return BWAHIP_ENODEV; }
	if (!g_rccl.load()) return BWAHIP_ENODEV;
	HIP_TRY(hipSetDevice(device));
	int rc = 0;
	Comm comm = nullptr;
	hipStream_t st = nullptr;
	bwahip_ctx *c = new bwahip_ctx();
	c->device = device; c->index_resident = true;
	HostIndex full;                                             // rank 0: the loaded files (FM-index arrays freed after the upload)
	memset(&full, 0, sizeof full);
	std::vector<uint8_t> meta;
	DevBuf d_meta;
	uint64_t meta_len = 0;
	bool collective_failure = false;                           // every rank leaves together: the communicator can be destroyed normally
	UniqueId id;
	memcpy(&id, id128, 128);
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	// rank 0 reads and packs the index BEFORE the communicator exists: whatever can fail on one rank alone (files, host memory) has then
	// either happened or not, and the first thing broadcast is rank 0's verdict -- the metadata length, 0 = "rank 0 failed" -- so that
	// every rank returns BWAHIP_EIO together instead of waiting in a broadcast that never comes
	if (rank == 0) {
		const int lrc = bwahip_load_index_files(prefix, &full);
		if (!lrc) { meta = pack_meta(full); meta_len = meta.size(); }
		else { fprintf(stderr, "[bwahip] rank 0 could not load %s: the other ranks are told\n", prefix); meta_len = 0; }
	}
	RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
	// 1. metadata: length (8 bytes; 0: rank 0 has no index), then the blob
	if ((rc = d_meta.ensure(64))) goto done;
	if (rank == 0 && hipMemcpyAsync(d_meta.p, &meta_len, 8, hipMemcpyHostToDevice, st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	RCCL_TRY(g_rccl.Broadcast(d_meta.p, d_meta.p, 8, NCCL_UINT8, 0, comm, st));
	if (hipMemcpyAsync(&meta_len, d_meta.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	if (meta_len == 0) { rc = BWAHIP_EIO; collective_failure = true; goto done; }      // agreed by all ranks: a clean shutdown of the communicator
	if (meta_len < sizeof(MetaHdr) || meta_len > (1ull << 32)) { rc = BWAHIP_EIO; goto done; }
	if ((rc = d_meta.ensure(meta_len))) goto done;
	if (rank == 0 && hipMemcpyAsync(d_meta.p, meta.data(), meta_len, hipMemcpyHostToDevice, st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	RCCL_TRY(g_rccl.Broadcast(d_meta.p, d_meta.p, meta_len, NCCL_UINT8, 0, comm, st));
	if (rank != 0) {
		meta.resize(meta_len);
		if (hipMemcpyAsync(meta.data(), d_meta.p, meta_len, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	}
	if (!unpack_meta(meta, &c->host)) { rc = BWAHIP_EIO; goto done; }
	// 2. the three arrays, into the context's own buffers
	{
		const size_t n_bwt = (size_t)c->host.bwt.bwt_size * 4, n_sa = (size_t)c->host.bwt.n_sa * 8, n_pac = (size_t)c->host.bns.l_pac / 4 + 1;
		if ((rc = c->d_bwt.ensure(n_bwt)) || (rc = c->d_sa.ensure(n_sa)) || (rc = c->d_pac.ensure(n_pac))) goto done;
		if (rank == 0) {
			if (hipMemcpyAsync(c->d_bwt.p, full.bwt.bwt, n_bwt, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(c->d_sa.p, full.bwt.sa, n_sa, hipMemcpyHostToDevice, st) != hipSuccess ||
			    hipMemcpyAsync(c->d_pac.p, full.pac, n_pac, hipMemcpyHostToDevice, st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
		}
		// in pieces of 1 GiB: the element count of one call stays far below 2^31 whatever the library's internal index type is
		for (int arr = 0; arr < 3; ++arr) {
			uint8_t *p = (uint8_t*)(arr == 0 ? c->d_bwt.p : arr == 1 ? c->d_sa.p : c->d_pac.p);
			const size_t total = arr == 0 ? n_bwt : arr == 1 ? n_sa : n_pac;
			for (size_t o = 0; o < total; o += (size_t)1 << 30) {
				const size_t len = total - o < ((size_t)1 << 30) ? total - o : (size_t)1 << 30;
				RCCL_TRY(g_rccl.Broadcast(p + o, p + o, len, NCCL_UINT8, 0, comm, st));
			}
		}
		// host copy of the packed reference (finalisation on host threads, when that knob is used)
		c->host.pac = (uint8_t*)malloc(n_pac);
		if (!c->host.pac) { rc = BWAHIP_ENOMEM; goto done; }
		if (hipMemcpyAsync(c->host.pac, c->d_pac.p, n_pac, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = BWAHIP_ENODEV; goto done; }
	}
	rc = ctx_setup(c, &c->host.bwt, &c->host.bns, c->host.pac);
done:
	d_meta.release();
	bwahip_free_host_index(&full);
	// a failure of THIS rank alone after the communicator exists would leave the others waiting in their next broadcast: abort the
	// communicator (its peers' pending operations then fail) instead of destroying it quietly
	if (comm) { if (rc && !collective_failure && g_rccl.CommAbort) g_rccl.CommAbort(comm); else g_rccl.CommDestroy(comm); }
	if (st) (void)hipStreamDestroy(st);
	if (rc) { bwahip_destroy(c); return rc; }
	*out = c;
	return 0;
}
