// K3 -- chaining and chain filtering (mem_chain body bwamem.c:272-317, mem_chain_flt bwamem.c:334-392).
//
// Both steps are strictly sequential per read and order-sensitive (greedy merge into the chain with the
// greatest pos <= rbeg, found through a B-tree whose shape decides where duplicate positions land,
// kbtree.h; then an UNSTABLE introsort by weight, ksort.h:176-227), so the sequential algorithms are
// restated literally.  k_chain runs one read per lane (ordinary reads: a handful of seeds); reads with
// hundreds to thousands of seeds go to k_chain_big, where a whole wavefront drives the same B-tree in LDS
// (they would otherwise set the duration of the launch), and the O(n^2) overlap filter of many-chain reads
// runs in k_chain_flt.  Working storage is per-read global memory sized from the seed count (no in-kernel
// allocation, no local arrays: nothing goes to scratch).
//   in : seeds of the read in look-up order (K2), sorted intervals (K1) for l_rep
//   out: filtered chains in final order with their seeds contiguous; optional pre-filter dump
#include "bwahip_internal.h"
#include "regsort_dev.h"

namespace {
using wv::RegKey; using wv::RegSort;

constexpr int BT_T = 6, BT_MAXK = 2 * BT_T - 1;              // kbtree.h:59 with sizeof(mem_chain_t) = 32
constexpr int FLT_SEQ_MAX = 32;                              // more chains than this: filter in k_chain_flt

// every key carries a copy of its chain's pos so that the search inside a node reads the node only: one memory
// latency per level instead of one per binary-search probe (the probes chased cw[key].pos through HBM/L2)
struct BtNode { int is_internal, n; int key[BT_MAXK]; int ptr[BT_MAXK + 1]; int pad; int64_t pos[BT_MAXK]; };
static_assert(sizeof(BtNode) == sizeof(BtNodeOpaque), "BtNode layout");
// chain under construction: seeds as a linked list; the query/reference ends test_and_merge and the filter look at
// are cached here (first seed: qbeg, rbeg == pos; last seed: qbeg, len, rbeg) to save a dependent seed fetch
struct ChainW { int64_t pos; int head, tail, n, rid; int64_t last_rbeg; int first_qbeg, last_qbeg, last_len, pad; };
static_assert(sizeof(ChainW) == sizeof(ChainWOpaque), "ChainW layout");

struct ReadCtx {
	const DevSeed *seeds; int n_seeds;
	ChainW *cw; int n_chains;
	int *nxt;
	BtNode *nodes; int n_nodes, root;
	bool coop; int lane;                                        // k_chain_big: the 64 lanes run the read together (uniform control flow)
	bool glb;                                                   // ... with the nodes in global memory (no LDS variant holds the read)
	int *ord, *wts, *kept, *first, *keep_list, *stack;
};

__device__ __forceinline__ int cmp_pos(int64_t a, int64_t b) { return (b < a) - (a < b); }

// kbtree.h:119 __kb_getp_aux.  The binary search over sorted keys returns the number of keys below `pos`; counting
// them directly gives the same index without dependent probes or dynamic register indexing.
__device__ int bt_find(const ReadCtx &c, const BtNode *x, int64_t pos, int *r)
{
	const int n = x->n;
	if (n == 0) return -1;
	int begin = 0;
	int64_t pb = 0;
#pragma unroll
	for (int m = 0; m < BT_MAXK; ++m) {
		const int64_t pm = x->pos[m];
		begin += (m < n && pm < pos) ? 1 : 0;
	}
	if (begin == n) { *r = 1; return n - 1; }
#pragma unroll
	for (int m = 0; m < BT_MAXK; ++m) pb = m == begin ? x->pos[m] : pb;
	*r = cmp_pos(pos, pb);
	if (*r < 0) --begin;
	return begin;
}

// kbtree.h:152 kb_intervalp (lower bound only)
__device__ int bt_lower(const ReadCtx &c, int64_t pos)
{
	int xi = c.root, lower = -1;
	for (;;) {
		const BtNode *x = &c.nodes[xi];
		int r = 0, i = bt_find(c, x, pos, &r);
		if (i >= 0 && r == 0) return x->key[i];
		if (i >= 0) lower = x->key[i];
		if (!x->is_internal) return lower;
		xi = x->ptr[i + 1];
	}
}

__device__ int bt_new(ReadCtx &c, int is_internal)
{
	BtNode *z = &c.nodes[c.n_nodes];
	z->is_internal = is_internal; z->n = 0;
	return c.n_nodes++;
}

// kbtree.h:172 __kb_split
__device__ void bt_split(ReadCtx &c, int xi, int i, int yi)
{
	int zi = bt_new(c, c.nodes[yi].is_internal);
	BtNode *x = &c.nodes[xi], *y = &c.nodes[yi], *z = &c.nodes[zi];
	z->n = BT_T - 1;
	for (int k = 0; k < BT_T - 1; ++k) { z->key[k] = y->key[k + BT_T]; z->pos[k] = y->pos[k + BT_T]; }
	if (y->is_internal) for (int k = 0; k < BT_T; ++k) z->ptr[k] = y->ptr[k + BT_T];
	y->n = BT_T - 1;
	for (int k = x->n; k > i; --k) x->ptr[k + 1] = x->ptr[k];
	x->ptr[i + 1] = zi;
	for (int k = x->n - 1; k >= i; --k) { x->key[k + 1] = x->key[k]; x->pos[k + 1] = x->pos[k]; }
	x->key[i] = y->key[BT_T - 1]; x->pos[i] = y->pos[BT_T - 1];
	++x->n;
}

// kbtree.h:188-222 kb_putp (iterative descent)
__device__ void bt_put(ReadCtx &c, int k)
{
	const int64_t pos = c.cw[k].pos;
	if (c.nodes[c.root].n == BT_MAXK) {
		int s = bt_new(c, 1);
		c.nodes[s].ptr[0] = c.root;
		bt_split(c, s, 0, c.root);
		c.root = s;
	}
	int xi = c.root;
	for (;;) {
		BtNode *x = &c.nodes[xi];
		int r = 0;
		if (!x->is_internal) {
			int i = bt_find(c, x, pos, &r);
			for (int t = x->n - 1; t > i; --t) { x->key[t + 1] = x->key[t]; x->pos[t + 1] = x->pos[t]; }
			x->key[i + 1] = k; x->pos[i + 1] = pos;
			++x->n;
			return;
		}
		int i = bt_find(c, x, pos, &r) + 1;
		if (c.nodes[x->ptr[i]].n == BT_MAXK) {
			bt_split(c, xi, i, x->ptr[i]);
			x = &c.nodes[xi];
			if (pos > x->pos[i]) ++i;
		}
		xi = x->ptr[i];
	}
}


// ---- k_chain_big: the same B-tree, driven by a whole wavefront.  Control flow is uniform (every lane executes the read);
// inside a node the lanes split the work: lane m holds key m, so the search is one compare and a ballot, and the
// shifts of an insert / split are one element per lane.  Nodes live in LDS; lane 0 does the single-element stores.
__device__ __forceinline__ void wsync(bool global_nodes = false)
{
	// orders the LDS node accesses of the lanes of this wavefront; deliberately NOT a full fence: waiting for the
	// outstanding global stores (chain records, seed links) several times per seed was the cost of this kernel.
	// global_nodes (reads beyond the LDS variants: their nodes live in global memory): the full fence it is.
	if (global_nodes) { __threadfence_block(); __syncthreads(); return; }
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int64_t wread64(int64_t v, int src)
{
	return (int64_t)((uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src) | (uint64_t)(uint32_t)__shfl((int)((uint64_t)v >> 32), src) << 32);
}
__device__ __forceinline__ int wbt_find(const BtNode *x, int64_t pos, int l, int *r)
{
	const int n = x->n;
	if (n == 0) return -1;
	const int64_t pm = l < BT_MAXK ? x->pos[l] : 0;
	int begin = __popcll(__ballot(l < n && pm < pos));          // keys ascend: the lanes below `pos` are a prefix
	if (begin == n) { *r = 1; return n - 1; }
	*r = cmp_pos(pos, wread64(pm, begin));
	if (*r < 0) --begin;
	return begin;
}
__device__ int wbt_lower(const ReadCtx &c, int64_t pos)
{
	int xi = c.root, lower = -1;
	for (;;) {
		const BtNode *x = &c.nodes[xi];
		int r = 0, i = wbt_find(x, pos, c.lane, &r);
		if (i >= 0 && r == 0) return x->key[i];
		if (i >= 0) lower = x->key[i];
		if (!x->is_internal) return lower;
		xi = x->ptr[i + 1];
	}
}
__device__ int wbt_new(ReadCtx &c, int is_internal)
{
	if (c.lane == 0) { BtNode *z = &c.nodes[c.n_nodes]; z->is_internal = is_internal; z->n = 0; }
	return c.n_nodes++;
}
__device__ void wbt_split(ReadCtx &c, int xi, int i, int yi)
{
	const int l = c.lane;
	const int y_internal = c.nodes[yi].is_internal;
	const int zi = wbt_new(c, y_internal);
	BtNode *x = &c.nodes[xi], *y = &c.nodes[yi], *z = &c.nodes[zi];
	const int xn = x->n;
	int mk = 0, mp = 0, sk = 0, sptr = 0; int64_t mpos = 0, spos = 0;
	if (l < BT_T - 1) { mk = y->key[l + BT_T]; mpos = y->pos[l + BT_T]; }
	if (y_internal && l < BT_T) mp = y->ptr[l + BT_T];
	if (l > i && l <= xn) sptr = x->ptr[l];
	if (l >= i && l < xn) { sk = x->key[l]; spos = x->pos[l]; }
	const int upk = y->key[BT_T - 1]; const int64_t uppos = y->pos[BT_T - 1];
	wsync(c.glb);
	if (l < BT_T - 1) { z->key[l] = mk; z->pos[l] = mpos; }
	if (y_internal && l < BT_T) z->ptr[l] = mp;
	if (l > i && l <= xn) x->ptr[l + 1] = sptr;
	if (l >= i && l < xn) { x->key[l + 1] = sk; x->pos[l + 1] = spos; }
	wsync(c.glb);
	if (l == 0) { z->n = BT_T - 1; y->n = BT_T - 1; x->ptr[i + 1] = zi; x->key[i] = upk; x->pos[i] = uppos; x->n = xn + 1; }
	wsync(c.glb);
}
__device__ void wbt_put(ReadCtx &c, int k, int64_t pos)
{
	const int l = c.lane;
	if (c.nodes[c.root].n == BT_MAXK) {
		const int s = wbt_new(c, 1);
		if (l == 0) c.nodes[s].ptr[0] = c.root;
		wsync(c.glb);
		wbt_split(c, s, 0, c.root);
		c.root = s;
	}
	int xi = c.root;
	for (;;) {
		BtNode *x = &c.nodes[xi];
		int r = 0;
		if (!x->is_internal) {
			const int i = wbt_find(x, pos, l, &r), n = x->n;
			int sk = 0; int64_t spos = 0;
			if (l > i && l < n) { sk = x->key[l]; spos = x->pos[l]; }
			wsync(c.glb);
			if (l > i && l < n) { x->key[l + 1] = sk; x->pos[l + 1] = spos; }
			if (l == 0) { x->key[i + 1] = k; x->pos[i + 1] = pos; x->n = n + 1; }
			wsync(c.glb);
			return;
		}
		int i = wbt_find(x, pos, l, &r) + 1;
		const int child = x->ptr[i];
		if (c.nodes[child].n == BT_MAXK) {
			wbt_split(c, xi, i, child);
			if (pos > x->pos[i]) ++i;
		}
		xi = x->ptr[i];
	}
}

// kbtree.h:336 __kb_traverse: in-order walk with an explicit stack (node, next child) in global memory
__device__ int bt_inorder(const ReadCtx &c, int *out)
{
	int n_out = 0, sp = 0;
	int *stk = c.stack;                                         // pairs (node, i)
	stk[0] = c.root; stk[1] = 0;
	while (sp >= 0) {
		int xi = stk[2 * sp], i = stk[2 * sp + 1];
		const BtNode *x = &c.nodes[xi];
		if (x->is_internal) {
			if (i <= x->n) {
				if (i > 0) out[n_out++] = x->key[i - 1];
				stk[2 * sp + 1] = i + 1;
				++sp; stk[2 * sp] = x->ptr[i]; stk[2 * sp + 1] = 0;
			} else --sp;
		} else {
			for (int t = 0; t < x->n; ++t) out[n_out++] = x->key[t];
			--sp;
		}
	}
	return n_out;
}

// bwamem.c:197 test_and_merge
__device__ bool try_merge(ReadCtx &c, const DevOpt &opt, int64_t l_pac, int ci, int si, const DevSeed p)
{
	ChainW *ch = &c.cw[ci];
	const ChainW w = *ch;
	const int64_t first_rbeg = w.pos;                           // the chain's pos is its first seed's rbeg (bwamem.c:304)
	int64_t qend = w.last_qbeg + w.last_len, rend = w.last_rbeg + w.last_len;
	if (p.rid != w.rid) return false;
	if (p.qbeg >= w.first_qbeg && p.qbeg + p.len <= qend && p.rbeg >= first_rbeg && p.rbeg + p.len <= rend) return true;
	if ((w.last_rbeg < l_pac || first_rbeg < l_pac) && p.rbeg >= l_pac) return false;
	int64_t x = p.qbeg - w.last_qbeg, y = p.rbeg - w.last_rbeg;
	if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - w.last_len < opt.max_chain_gap && y - w.last_len < opt.max_chain_gap) {
		c.nxt[w.tail] = si; c.nxt[si] = -1;
		ch->tail = si; ch->n = w.n + 1; ch->last_rbeg = p.rbeg; ch->last_qbeg = p.qbeg; ch->last_len = p.len;
		return true;
	}
	return false;
}

// bwamem.c:220 mem_chain_weight
__device__ int chain_weight(const ReadCtx &c, int ci)
{
	int64_t end = 0;
	int w = 0, tmp;
	for (int s = c.cw[ci].head; s >= 0; s = c.nxt[s]) {
		const DevSeed d = c.seeds[s];
		if (d.qbeg >= end) w += d.len;
		else if (d.qbeg + d.len > end) w += (int)(d.qbeg + d.len - end);
		end = end > d.qbeg + d.len ? end : d.qbeg + d.len;
	}
	tmp = w; w = 0; end = 0;
	for (int s = c.cw[ci].head; s >= 0; s = c.nxt[s]) {
		const DevSeed d = c.seeds[s];
		if (d.rbeg >= end) w += d.len;
		else if (d.rbeg + d.len > end) w += (int)(d.rbeg + d.len - end);
		end = end > d.rbeg + d.len ? end : d.rbeg + d.len;
	}
	w = w < tmp ? w : tmp;
	return w < 1 << 30 ? w : (1 << 30) - 1;
}

// ---- ksort.h:146-227 on an array of chain indices, key = weight, "less" = heavier first (bwamem.c:331)
#define W_LT(a, b) (c.wts[(a)] > c.wts[(b)])
__device__ void isort_insertion(const ReadCtx &c, int *s, int *t)
{
	for (int *i = s + 1; i < t; ++i)
		for (int *j = i; j > s && W_LT(*j, *(j - 1)); --j) { int tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
__device__ void isort_comb(const ReadCtx &c, int n, int *a)
{
	const double shrink = 1.2473309501039786540366528676643;
	int swapped, gap = n;
	do {
		if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = 0;
		for (int *i = a; i < a + n - gap; ++i) {
			int *j = i + gap;
			if (W_LT(*j, *i)) { int tmp = *i; *i = *j; *j = tmp; swapped = 1; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) isort_insertion(c, a, a + n);
}
__device__ void isort_weight(const ReadCtx &c, int n, int *a)
{
	int d, top = 0, *stk = c.stack;                             // frames (lo, hi, depth)
	int *s, *t, *i, *j, *k, pivot, tmp;
	if (n < 1) return;
	if (n == 2) { if (W_LT(a[1], a[0])) { tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return; }
	for (d = 2; 1 << d < n; ++d);
	s = a; t = a + (n - 1); d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { isort_comb(c, (int)(t - s) + 1, s); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (W_LT(*k, *i)) { if (W_LT(*k, *j)) k = j; }
			else k = W_LT(*j, *i) ? i : j;
			pivot = *k;
			if (k != t) { tmp = *k; *k = *t; *t = tmp; }
			for (;;) {
				do ++i; while (i < t && W_LT(*i, pivot));          // i stops at the pivot (at t) at the latest
				do --j; while (i <= j && W_LT(pivot, *j));
				if (j <= i) break;
				tmp = *i; *i = *j; *j = tmp;
			}
			tmp = *i; *i = *t; *t = tmp;
			if (i - s > t - i) {
				if (i - s > 16) { stk[3*top] = (int)(s - a); stk[3*top+1] = (int)(i - 1 - a); stk[3*top+2] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stk[3*top] = (int)(i + 1 - a); stk[3*top+1] = (int)(t - a); stk[3*top+2] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { isort_insertion(c, a, a + n); return; }
			--top; s = a + stk[3*top]; t = a + stk[3*top+1]; d = stk[3*top+2];
		}
	}
}
#undef W_LT

__device__ __forceinline__ int chn_beg(const ReadCtx &c, int ci) { return c.cw[ci].first_qbeg; }
__device__ __forceinline__ int chn_end(const ReadCtx &c, int ci) { return c.cw[ci].last_qbeg + c.cw[ci].last_len; }

__device__ void write_chains(const ReadCtx &c, const DevIndex &ix, int n, const int *order, float frac_rep, bool with_flt,
                             DevChain *oc, DevSeed *os)
{
	int so = 0;
	for (int k = 0; k < n; ++k) {
		int ci = order[k];
		DevChain h;
		h.pos = c.cw[ci].pos; h.seed_off = so; h.n = c.cw[ci].n; h.rid = c.cw[ci].rid;
		h.w = with_flt ? c.wts[ci] : 0; h.kept = with_flt ? c.kept[ci] : 0; h.first = with_flt ? c.first[ci] : 0;
		h.is_alt = ix.anns[h.rid].is_alt ? 1 : 0; h.frac_rep = frac_rep;
		oc[k] = h;
		for (int s = c.cw[ci].head; s >= 0; s = c.nxt[s]) os[so++] = c.seeds[s];
	}
}

// BIG = false: one read per lane.  Reads with big_min < seeds <= big_max are only listed; k_chain<true> then takes one
// of them per workgroup (one working lane) with the B-tree nodes in LDS: the build is a chain of dependent node
// visits, ~10 per seed, and a 2 000-seed read otherwise sets the duration of the whole launch through L2 latency.
constexpr int BIG_NODES = 800;                               // 800 x 192 B = 150 KB of LDS; nodes <= 0.24 x seeds + a few
constexpr int MID_NODES = 400, MID_SEEDS = 1536;             // two workgroups per CU for the (far more common) reads up to 1536 seeds
constexpr int SMALL_NODES = 72, SMALL_SEEDS = 256;           // eleven per CU for reads of a few dozen to 256 seeds (14 KB)
constexpr int GLB_LDS = 2048 + 512;                          // reads beyond big_max seeds: nodes in their global slots, this much LDS for the sort's small tables (+ the unused ends)
// classes of the wavefront-per-read kernels: 0 = up to SMALL_SEEDS, 1 = up to MID_SEEDS, 2 = up to big_max, 3 = beyond (global nodes)
__device__ __forceinline__ int coop_class(int S, int mid_max, int big_max) { return S <= SMALL_SEEDS ? 0 : S <= mid_max ? 1 : S <= big_max ? 2 : 3; }

template <bool BIG>
__device__ __forceinline__ void chain_read(const ChainLaunch &a, int r, BtNode *lds_nodes, int lds_node_cap = BIG_NODES, uint8_t *lds_area = nullptr, int lds_bytes = 0)
{
	const int64_t sb = a.seed_base[r];
	const int S = (int)(a.seed_base[r + 1] - sb), len = (int)(a.off[r + 1] - a.off[r]);
	if (!BIG && a.big_list && S > a.big_min) return;               // a wavefront-per-read kernel's (listed by k_chain_classify); they run concurrently: touch nothing
	ReadCtx c;
	c.seeds = a.seeds + sb; c.n_seeds = S;
	c.cw = reinterpret_cast<ChainW*>(a.cw_) + sb; c.nxt = a.nxt + sb; c.ord = a.ord + sb; c.wts = a.wts + sb; c.kept = a.kept + sb; c.first = a.first + sb;
	c.keep_list = a.keep_list + sb;
	c.nodes = BIG && lds_nodes ? lds_nodes : reinterpret_cast<BtNode*>(a.nodes_) + (sb >> 2) + 4 * (int64_t)r;   // (class 3: the read's global node slots)
	c.stack = a.stack + 256 * (int64_t)r;
	c.n_chains = 0; c.n_nodes = 0; c.coop = BIG; c.lane = (int)(threadIdx.x & 63); c.glb = BIG && !lds_nodes;
	a.chain_n[r] = 0; a.kept_seeds[r] = 0;
	if (a.dbg_chain_n) a.dbg_chain_n[r] = 0;
	if (S == 0) return;

	// frac_rep: union length of the query spans of over-abundant intervals (bwamem.c:272-279)
	int l_rep = 0;
	{
		const DevIntv *iv = a.intv + (size_t)r * a.cap;
		int n = a.intv_n[r], b = 0, e = 0;
		for (int t = 0; t < n; ++t) {
			if (iv[t].x2 <= (uint64_t)a.opt.max_occ) continue;
			int sbq = (int)(iv[t].info >> 32), seq = (int)(uint32_t)iv[t].info;
			if (sbq > e) { l_rep += e - b; b = sbq; e = seq; }
			else e = e > seq ? e : seq;
		}
		l_rep += e - b;
	}
	const float frac_rep = (float)l_rep / len;                  // bwamem.c:317

	const unsigned long long t_0 = wall_clock64();
	// greedy chaining (bwamem.c:280-308)
	c.root = BIG ? wbt_new(c, 0) : bt_new(c, 0);
	if (BIG) wsync(c.glb);
	DevSeed sd_next = c.seeds[0];                               // the next seed is fetched one iteration ahead of its use
	for (int si = 0; si < S; ++si) {
		const DevSeed sd = sd_next;
		if (si + 1 < S) sd_next = c.seeds[si + 1];
		if (sd.rid < 0) continue;                               // bwamem.c:294
		bool to_add = true;
		if (c.n_chains) {
			int lower = BIG ? wbt_lower(c, sd.rbeg) : bt_lower(c, sd.rbeg);
			if (lower >= 0 && try_merge(c, a.opt, a.ix.l_pac, lower, si, sd)) to_add = false;
		}
		if (to_add) {
			ChainW *ch = &c.cw[c.n_chains];
			ch->pos = sd.rbeg; ch->head = ch->tail = si; ch->n = 1; ch->rid = sd.rid;
			ch->last_rbeg = sd.rbeg; ch->first_qbeg = ch->last_qbeg = sd.qbeg; ch->last_len = sd.len; ch->pad = 0;
			c.nxt[si] = -1;
			if (BIG) { wsync(c.glb); wbt_put(c, c.n_chains++, sd.rbeg); } else bt_put(c, c.n_chains++);
		}
	}
	const unsigned long long t_1 = wall_clock64();
	int n_chn = bt_inorder(c, c.ord);                           // bwamem.c:311-315
	if (a.dbg_chain_n) {                                        // stage dump: chains before filtering
		a.dbg_chain_n[r] = n_chn;
		write_chains(c, a.ix, n_chn, c.ord, frac_rep, false, a.dbg_chains + sb, a.dbg_seeds + sb);
	}

	// ---- mem_chain_flt (bwamem.c:334-392)
	if (n_chn == 0) return;
	int k = 0;
	if (BIG) {                                                  // weights: one chain per lane (each walks its own seed list)
		for (int i = c.lane; i < n_chn; i += 64) {
			const int ci = c.ord[i];
			c.first[ci] = -1; c.kept[ci] = 0;
			c.wts[ci] = chain_weight(c, ci);
		}
		__threadfence_block(); __builtin_amdgcn_wave_barrier();
		for (int i = 0; i < n_chn; ++i) { const int ci = c.ord[i]; if (c.wts[ci] >= a.opt.min_chain_weight) c.ord[k++] = ci; }
	} else {
		for (int i = 0; i < n_chn; ++i) {
			int ci = c.ord[i];
			c.first[ci] = -1; c.kept[ci] = 0;
			c.wts[ci] = chain_weight(c, ci);
			if (c.wts[ci] >= a.opt.min_chain_weight) c.ord[k++] = ci;
		}
	}
	n_chn = k;
	if (n_chn == 0) return;
	if (BIG) {
		// ks_introsort by weight (bwamem.c:331-332,348), heavier first: the whole wavefront follows the reference's introsort exactly, ties
		// included (isort_dev.h); the B-tree is no longer needed, so its LDS (or, for the reads with global nodes, the filter's scratch of
		// the read, free until the filter) holds the keys, the index array and the sort's tables.  The one-lane restatement takes over only
		// at the introsort's depth limit.
		// (LDS is reached through generic pointers here: 256 bytes at either end of the array are left alone, so that no folded offset
		// can put a base register outside the LDS aperture -- DESIGN 4.2)
		uint8_t *area = lds_nodes ? lds_area + 256 : reinterpret_cast<uint8_t*>(a.flt + 8 * sb);
		const size_t area_bytes = lds_nodes ? (size_t)lds_bytes - 512 : (size_t)S * 32;
		// layout: keys[n] (16 B) | idx[n] | work (8-byte aligned) ; the stack and the 256-word table always in LDS
		int *stk = reinterpret_cast<int*>(lds_nodes ? lds_area + lds_bytes - 256 - 2048 : lds_area + 256);
		unsigned *tab = reinterpret_cast<unsigned*>(stk + 256);
		const size_t need = (size_t)n_chn * 20 + 8 + wv::ws_work_ints(n_chn) * 4 + (lds_nodes ? 2048 : 0);
		bool sorted = false;
		if (need <= area_bytes && n_chn >= 2) {
			RegKey *keys = reinterpret_cast<RegKey*>(area);
			int *idx = reinterpret_cast<int*>(keys + n_chn);
			int *work = idx + n_chn + (n_chn & 1);
			for (int i = c.lane; i < n_chn; i += 64) { keys[i].k64 = -(int64_t)c.wts[c.ord[i]]; keys[i].score = 0; keys[i].qb = 0; idx[i] = i; }
			__threadfence_block(); __syncthreads();
			sorted = wv::wave_sort_exact(RegSort{keys, 0}, n_chn, idx, work, stk, tab, c.lane);
			__threadfence_block(); __syncthreads();
			if (sorted) {                                           // positions -> chain ids, through the (free again) work area
				for (int i = c.lane; i < n_chn; i += 64) work[i] = c.ord[idx[i]];
				__threadfence_block(); __syncthreads();
				for (int i = c.lane; i < n_chn; i += 64) c.ord[i] = work[i];
				__threadfence_block(); __syncthreads();
			}
		}
		if (!sorted && n_chn >= 2) {
			if (c.lane == 0) isort_weight(c, n_chn, c.ord);
			__threadfence_block(); __syncthreads();
		}
	} else isort_weight(c, n_chn, c.ord);                       // exact unstable introsort
	const unsigned long long t_2 = wall_clock64();
	if (n_chn > FLT_SEQ_MAX && a.heavy_list) {
		// many chains: the O(n^2) overlap filter runs wavefront-parallel in k_chain_flt; leave it the per-position data
		int *fb = a.flt + 8 * sb, *fe = fb + S, *fw = fe + S;
		for (int i = 0; i < n_chn; ++i) {
			const int ci = c.ord[i];
			fb[i] = chn_beg(c, ci) | (a.ix.anns[c.cw[ci].rid].is_alt ? 1 << 30 : 0);
			fe[i] = chn_end(c, ci);
			fw[i] = c.wts[ci];
		}
		a.chain_n[r] = -n_chn;                                  // pending marker
		if (!BIG || c.lane == 0) a.heavy_list[atomicAdd(a.heavy_count, 1)] = r;
		if (a.counters) { atomicMax(&cnt_row(a.counters)[8], t_1 - t_0); atomicMax(&cnt_row(a.counters)[9], t_2 - t_1); atomicMax(&cnt_row(a.counters)[12], (unsigned long long)S); atomicMax(&cnt_row(a.counters)[13], (unsigned long long)n_chn); }
		return;
	}
	// NB: `first` and the kept list hold positions in the sorted array, as in the reference
	int n_keep = 0;
	c.kept[c.ord[0]] = 3;
	c.keep_list[n_keep++] = 0;
	for (int i = 1; i < n_chn; ++i) {
		const int ai = c.ord[i];
		int large_ovlp = 0, kk;
		for (kk = 0; kk < n_keep; ++kk) {
			const int j = c.keep_list[kk], aj = c.ord[j];
			int bi = chn_beg(c, ai), bj = chn_beg(c, aj), ei = chn_end(c, ai), ej = chn_end(c, aj);
			int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
			bool j_alt = a.ix.anns[c.cw[aj].rid].is_alt != 0, i_alt = a.ix.anns[c.cw[ai].rid].is_alt != 0;
			if (e_min > b_max && (!j_alt || i_alt)) {
				int li = ei - bi, lj = ej - bj, min_l = li < lj ? li : lj;
				if ((float)(e_min - b_max) >= (float)min_l * a.opt.mask_level && min_l < a.opt.max_chain_gap) {
					large_ovlp = 1;
					if (c.first[aj] < 0) c.first[aj] = i;
					if ((float)c.wts[ai] < (float)c.wts[aj] * a.opt.drop_ratio && c.wts[aj] - c.wts[ai] >= a.opt.min_seed_len << 1) break;
				}
			}
		}
		if (kk == n_keep) { c.keep_list[n_keep++] = i; c.kept[ai] = large_ovlp ? 2 : 3; }
	}
	for (int i = 0; i < n_keep; ++i) {
		int ci = c.ord[c.keep_list[i]];
		if (c.first[ci] >= 0) c.kept[c.ord[c.first[ci]]] = 1;
	}
	{
		int i;
		for (i = k = 0; i < n_chn; ++i) {                       // bwamem.c:380-385
			int kp = c.kept[c.ord[i]];
			if (kp == 0 || kp == 3) continue;
			if (++k >= a.opt.max_chain_extend) break;
		}
		for (; i < n_chn; ++i) if (c.kept[c.ord[i]] < 3) c.kept[c.ord[i]] = 0;
	}
	int n_out = 0, tot = 0;
	for (int i = 0; i < n_chn; ++i) {
		int ci = c.ord[i];
		if (c.kept[ci] != 0) { c.ord[n_out++] = ci; tot += c.cw[ci].n; }
	}
	const unsigned long long t_3 = wall_clock64();
	write_chains(c, a.ix, n_out, c.ord, frac_rep, true, a.chains + sb, a.chain_seeds + sb);
	a.chain_n[r] = n_out;
	a.kept_seeds[r] = tot;
	if (a.counters) {                                           // phase maxima in 10 ns ticks (profiling aid)
		atomicMax(&cnt_row(a.counters)[8], t_1 - t_0); atomicMax(&cnt_row(a.counters)[9], t_2 - t_1); atomicMax(&cnt_row(a.counters)[10], t_3 - t_2);
		atomicMax(&cnt_row(a.counters)[11], wall_clock64() - t_3); atomicMax(&cnt_row(a.counters)[12], (unsigned long long)S); atomicMax(&cnt_row(a.counters)[13], (unsigned long long)n_chn);
	}
}

__global__ __launch_bounds__(64) void k_chain(ChainLaunch a)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	// one read per lane, in input order: packing the reads with many seeds into the same wavefronts was tried (launch order by seed count) and lost
	// -- 64 lanes each walking its own B-tree of hundreds of chains evict each other from the caches (k_chain 8.9 -> 13.2 ms per million reads)
	if (t < a.n_reads) chain_read<false>(a, t, nullptr);
}

// the reads the wavefront-per-read kernels take (so that they can run beside k_chain on other streams): four lists, one per class, in
// big_list[class * n_reads ..]; big_count[class] = length, big_count[4 + class] = the class's work-queue head
__global__ void k_chain_classify(ChainLaunch a)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= a.n_reads) return;
	const int S = (int)(a.seed_base[r + 1] - a.seed_base[r]);
	if (S > a.big_min) {
		const int cls = coop_class(S, a.mid_max, a.big_max);
		a.big_list[(size_t)cls * a.n_reads + atomicAdd(&a.big_count[cls], 1)] = r;
	}
}

template <int NODES, int CLS>
__global__ __launch_bounds__(64) void k_chain_big(ChainLaunch a)
{
	constexpr int LDS_BYTES = NODES > 0 ? NODES * (int)sizeof(BtNode) : GLB_LDS;
	__shared__ __attribute__((aligned(16))) uint8_t s_area[LDS_BYTES];
	BtNode *nodes = NODES > 0 ? reinterpret_cast<BtNode*>(s_area) : nullptr;
	const int n_mine = a.big_count[CLS];
	for (;;) {                                                   // work queue: a workgroup that finishes takes the next read
		int h = 0;
		if ((threadIdx.x & 63) == 0) h = atomicAdd(&a.big_count[4 + CLS], 1);
		h = __shfl(h, 0);
		if (h >= n_mine) break;
		chain_read<true>(a, a.big_list[(size_t)CLS * a.n_reads + h], nodes, NODES, s_area, LDS_BYTES);
		__threadfence_block(); __syncthreads();
	}
	// The kernel with the nodes in global memory is slower per read but is not held to one or two workgroups per CU by 77 / 150 KB of LDS.
	// When a batch has thousands of reads for the LDS kernels (a human-like repeat load: they alone took 0.25 s per million reads), its
	// workgroups help out: a queue that began with more reads than the LDS kernel's workgroups take in two rounds is emptied together with
	// that kernel (with a few reads in the queue -- the ordinary batch -- the faster kernels keep them all).
	if (CLS == 3) {
		for (int cls = 2; cls >= 1; --cls) {
			const int n_cls = a.big_count[cls];
			if (n_cls <= (cls == 2 ? 2 * 256 : 2 * 1024)) continue;
			for (;;) {
				int h = n_cls;
				if ((threadIdx.x & 63) == 0) h = atomicAdd(&a.big_count[4 + cls], 1);
				h = __shfl(h, 0);
				if (h >= n_cls) break;
				chain_read<true>(a, a.big_list[(size_t)cls * a.n_reads + h], nullptr, 0, s_area, LDS_BYTES);
				__threadfence_block(); __syncthreads();
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------
// k_chain_flt: the pairwise overlap filter of mem_chain_flt (bwamem.c:350-392) for reads with many chains,
// one read per wavefront.  For chain i (sequential, heavier first) the kept chains are tested 64 at a time;
// the reference's sequential semantics -- stop at the first kept chain that shadows i, having set `first`
// on every significantly overlapping kept chain up to and including it -- are recovered with ballots.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_chain_flt(ChainLaunch a)
{
	const int l = (int)(threadIdx.x & 63);
	const int n_heavy = *a.heavy_count;
	for (int hi = blockIdx.x; hi < n_heavy; hi += gridDim.x) {
		const int r = a.heavy_list[hi];
		const int64_t sb = a.seed_base[r];
		const int S = (int)(a.seed_base[r + 1] - sb), len = (int)(a.off[r + 1] - a.off[r]);
		const int n = -a.chain_n[r];
		const unsigned long long t_0 = wall_clock64();
		int *ord = a.ord + sb, *nxt = a.nxt + sb;
		ChainW *cw = reinterpret_cast<ChainW*>(a.cw_) + sb;
		const DevSeed *seeds = a.seeds + sb;
		int *fb = a.flt + 8 * sb, *fe = fb + S, *fw = fe + S, *Kb = fw + S, *Ke = Kb + S, *Kw = Ke + S, *ooff = Kw + S, *oidx = ooff + S;
		int *kept = a.kept + sb, *Kfirst = a.first + sb;         // kept: by sorted position; Kfirst: by kept index
		int *kmap = a.keep_list + sb;                            // sorted position -> kept index (-1: not in the kept list)
		for (int i = l; i < n; i += 64) { kept[i] = 0; kmap[i] = -1; }
		__threadfence_block(); __syncthreads();
		int nk = 0;
		if (l == 0) { kept[0] = 3; kmap[0] = 0; Kb[0] = fb[0]; Ke[0] = fe[0]; Kw[0] = fw[0]; Kfirst[0] = -1; }
		nk = 1;
		__threadfence_block(); __syncthreads();
		for (int i = 1; i < n; ++i) {
			const int bi_ = fb[i], ei = fe[i], wi = fw[i];
			const int bi = bi_ & 0x3fffffff; const bool i_alt = (bi_ >> 30) & 1;
			bool large = false, broke = false;
			for (int base = 0; base < nk && !broke; base += 64) {
				const int kk = base + l;
				bool sig = false, brk = false;
				if (kk < nk) {
					const int bj_ = Kb[kk], ej = Ke[kk], wj = Kw[kk];
					const int bj = bj_ & 0x3fffffff; const bool j_alt = (bj_ >> 30) & 1;
					const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
					if (e_min > b_max && (!j_alt || i_alt)) {
						const int li = ei - bi, lj = ej - bj, min_l = li < lj ? li : lj;
						if ((float)(e_min - b_max) >= (float)min_l * a.opt.mask_level && min_l < a.opt.max_chain_gap) {
							sig = true;
							if ((float)wi < (float)wj * a.opt.drop_ratio && wj - wi >= a.opt.min_seed_len << 1) brk = true;
						}
					}
				}
				const unsigned long long m_brk = __ballot(brk), m_sig = __ballot(sig);
				int upto = 64;                                  // lanes < upto take part
				if (m_brk) { upto = __ffsll((long long)m_brk); broke = true; }   // includes the breaking lane
				const unsigned long long in = upto >= 64 ? ~0ull : ((1ull << upto) - 1);
				if (m_sig & in) large = true;
				if (sig && l < upto && Kfirst[kk] < 0) Kfirst[kk] = i;
			}
			if (!broke) {
				if (l == 0) { kept[i] = large ? 2 : 3; kmap[i] = nk; Kb[nk] = bi_; Ke[nk] = ei; Kw[nk] = wi; Kfirst[nk] = -1; }
				++nk;
			}
			__threadfence_block(); __syncthreads();
		}
		for (int kk = l; kk < nk; kk += 64) { const int f = Kfirst[kk]; if (f >= 0) kept[f] = 1; }   // bwamem.c:376-379
		__threadfence_block(); __syncthreads();
		if (l == 0) {
			int i, k = 0;
			if (a.opt.max_chain_extend < n) {                     // bwamem.c:380-385 (no-op at the default 1<<30)
				for (i = 0; i < n; ++i) { const int kp = kept[i]; if (kp == 0 || kp == 3) continue; if (++k >= a.opt.max_chain_extend) break; }
				for (; i < n; ++i) if (kept[i] < 3) kept[i] = 0;
			}
			int n_out = 0, tot = 0;
			for (i = 0; i < n; ++i) if (kept[i] != 0) { oidx[n_out] = i; ooff[n_out] = tot; tot += cw[ord[i]].n; ++n_out; }
			a.chain_n[r] = n_out; a.kept_seeds[r] = tot;
		}
		__threadfence_block(); __syncthreads();
		const int n_out = a.chain_n[r];
		// frac_rep (bwamem.c:272-279, 317)
		int l_rep = 0;
		{
			const DevIntv *iv = a.intv + (size_t)r * a.cap;
			int ni = a.intv_n[r], b = 0, e = 0;
			for (int t = 0; t < ni; ++t) {
				if (iv[t].x2 <= (uint64_t)a.opt.max_occ) continue;
				int sbq = (int)(iv[t].info >> 32), seq = (int)(uint32_t)iv[t].info;
				if (sbq > e) { l_rep += e - b; b = sbq; e = seq; }
				else e = e > seq ? e : seq;
			}
			l_rep += e - b;
		}
		const float frac_rep = (float)l_rep / len;
		for (int k = l; k < n_out; k += 64) {                     // one chain per lane: header + seed copy
			const int i = oidx[k], ci = ord[i];
			DevChain h;
			h.pos = cw[ci].pos; h.seed_off = ooff[k]; h.n = cw[ci].n; h.rid = cw[ci].rid;
			h.w = fw[i]; h.kept = kept[i]; h.first = kmap[i] >= 0 ? Kfirst[kmap[i]] : -1; h.is_alt = (fb[i] >> 30) & 1; h.frac_rep = frac_rep;
			a.chains[sb + k] = h;
			int so = ooff[k];
			for (int s = cw[ci].head; s >= 0; s = nxt[s]) a.chain_seeds[sb + so++] = seeds[s];
		}
		if (l == 0 && a.counters) atomicMax(&cnt_row(a.counters)[10], wall_clock64() - t_0);
		__syncthreads();
	}
}

} // namespace

int launch_chain_flt(const ChainLaunch &a, hipStream_t st)
{
	if (a.n_reads <= 0 || !a.heavy_list) return 0;
	hipLaunchKernelGGL(k_chain_flt, dim3(4096), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_chain(const ChainLaunch &a, hipStream_t st, hipStream_t st2, hipStream_t st3, hipEvent_t fork, hipEvent_t join, hipEvent_t join3)
{
	if (a.n_reads <= 0) return 0;
	if (a.big_list) {                                        // reads with more than big_min seeds: a wavefront each, in kernels that run beside k_chain
		hipLaunchKernelGGL(k_chain_classify, dim3((a.n_reads + 255) / 256), dim3(256), 0, st, a);
		if (hipEventRecord(fork, st) != hipSuccess || hipStreamWaitEvent(st2, fork, 0) != hipSuccess || hipStreamWaitEvent(st3, fork, 0) != hipSuccess) return BWAHIP_ENODEV;
		hipLaunchKernelGGL((k_chain_big<0, 3>), dim3(a.glb_grid), dim3(64), 0, st2, a);       // the longest first
		hipLaunchKernelGGL((k_chain_big<BIG_NODES, 2>), dim3(256), dim3(64), 0, st2, a);
		hipLaunchKernelGGL((k_chain_big<MID_NODES, 1>), dim3(1024), dim3(64), 0, st3, a);
		hipLaunchKernelGGL((k_chain_big<SMALL_NODES, 0>), dim3(256 * 11), dim3(64), 0, st3, a);
		if (hipEventRecord(join, st2) != hipSuccess || hipEventRecord(join3, st3) != hipSuccess) return BWAHIP_ENODEV;
	}
	hipLaunchKernelGGL(k_chain, dim3((a.n_reads + 63) / 64), dim3(64), 0, st, a);
	if (a.big_list && (hipStreamWaitEvent(st, join, 0) != hipSuccess || hipStreamWaitEvent(st, join3, 0) != hipSuccess)) return BWAHIP_ENODEV;
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
