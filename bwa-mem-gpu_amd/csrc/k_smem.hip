// K1 -- SMEM / seed-interval collection (mem_collect_intv, bwamem.c:137-185) on gfx950.
//
// The algorithm is a long chain of dependent bwt_extend calls (bwt.c:262), each a random 64-byte
// gather (two when the interval straddles an Occ block).  A single read exposes almost no memory-level
// parallelism and ~1-2 us of loaded HBM latency per step must be hidden, so a wavefront carries many reads:
// a *group* of G lanes owns one read and runs it as an explicit state machine whose every transition
// issues exactly one bwt_extend.  All groups of a wavefront therefore reach the extend call site together
// and their gathers are in flight together.  G = 1 (default: one read per lane, the lane fetches and counts
// its own 64-byte blocks, 128 gathers in flight per wavefront-step), 2 (pair), 4 (quad: 16 B per lane of each
// block) or 8 (one quad per block); see fmi_dev.h.  Groups pull reads from a global work queue.  Reads that turn
// out to need thousands of steps are handed to k_smem_heavy (below), pass 3 runs in k_smem3, the sort in k_intv_sort.
//
// Per-read state machine (states cite the code they restate):
//   pass 1  bwamem.c:144-154   for x: bwt_smem1(x, min_intv=1)        FWD (bwt.c:304-320) then BWD (bwt.c:326-345)
//   pass 2  bwamem.c:156-165   re-seed long, rare SMEMs from their middle with min_intv = occ+1
//   pass 3  bwamem.c:167-182   bwt_seed_strategy1 (bwt.c:358): forward-only, stop at occ < max_mem_intv -- its own kernel,
//                              k_smem3 (independent of passes 1-2; a lean loop instead of one more state in this machine)
//   sort    bwamem.c:184       k_intv_sort: by info=(qbeg<<32|qend); entries with equal info are the same bi-interval
//                              (same query substring), so any sort yields the reference's array
// Lists prev/curr (bwt.c:293) are ONE in-place list of 16-byte packed entries per read: curr[k] (k <= j) overwrites
// the already consumed prev[k], so the swap of bwt.c:340 is a change of (base, n).  The 8*G entries at the top of the
// list -- the part the backward sweep works on -- live in an LDS ring (the memory system is request-rate bound on this
// access pattern -- see scripts/gather_bw.hip -- and the list was more than half of all requests); the bottom of longer
// lists is moved to a per-group HBM area as the forward pass overwrites the ring.  The per-call `mem`
// vector of bwt_smem1a is not materialised: only its last start coordinate is needed (bwt.c:333).
// Every lane of a group keeps an identical copy of the state; lane 0 of the group writes the LDS entries.
#include "fmi_dev.h"

namespace {

enum : int { ST_IDLE = 0, ST_NEXT, ST_FWD, ST_BWD, ST_FINISH };

__device__ __forceinline__ void put(DevIntv *p, uint64_t x0, uint64_t x1, uint64_t x2, uint64_t info)
{
	ulonglong2 *q = reinterpret_cast<ulonglong2*>(p);
	q[0] = make_ulonglong2(x0, x1);
	q[1] = make_ulonglong2(x2, info);
}
__device__ __forceinline__ void get(const DevIntv *p, uint64_t &x0, uint64_t &x1, uint64_t &x2, uint64_t &info)
{
	const ulonglong2 *q = reinterpret_cast<const ulonglong2*>(p);
	ulonglong2 a = q[0], b = q[1];
	x0 = a.x; x1 = a.y; x2 = b.x; info = b.y;
}

// NB: every access to list entries goes through put/get/get_info/get_x2 (one access path, one type):
// mixing these vector accesses with DevIntv member loads lets type-based alias analysis reorder them.
__device__ __forceinline__ uint64_t get_info(const DevIntv *p) { return reinterpret_cast<const ulonglong2*>(p)[1].y; }
__device__ __forceinline__ uint64_t get_x2(const DevIntv *p) { return reinterpret_cast<const ulonglong2*>(p)[1].x; }

// list entry = (x0, x1, x2 : 38 bits each, query end : 14 bits) = 16 bytes; runtime.hip refuses indexes >= 2^38 positions.
// Words 0-2 hold the low 32 bits of x0/x1/x2, word 3 the three 6-bit tops and the end: only 32-bit shifts to pack/unpack.
__device__ __forceinline__ uint4 pack_entry(uint64_t x0, uint64_t x1, uint64_t x2, uint32_t end)
{
	const uint32_t h0 = (uint32_t)(x0 >> 32), h1 = (uint32_t)(x1 >> 32), h2 = (uint32_t)(x2 >> 32);
	return make_uint4((uint32_t)x0, (uint32_t)x1, (uint32_t)x2, h0 | h1 << 6 | h2 << 12 | end << 18);
}
__device__ __forceinline__ void unpack_entry(uint4 v, uint64_t &x0, uint64_t &x1, uint64_t &x2, uint64_t &end)
{
	x0 = (uint64_t)(v.w & 63) << 32 | v.x;
	x1 = (uint64_t)(v.w >> 6 & 63) << 32 | v.y;
	x2 = (uint64_t)(v.w >> 12 & 63) << 32 | v.z;
	end = v.w >> 18;
}

// The read as 4-bit codes, 16 bases per 64-bit word (k_pack4 below); positions past the end hold 0xF.  A lane keeps
// the word it is walking in a register, so the per-step base look-up is a shift and only every 16th step is a load
// (byte loads of q[i] missed the L1 on nearly every step: the Occ gathers stream through it).
__device__ __forceinline__ int qbase(const uint64_t *row, int p, uint64_t &qw, int &qwi)
{
	const int wi = p >> 4;
	if (wi != qwi) { qw = row[wi]; qwi = wi; }
	const uint32_t half = (p & 8) ? (uint32_t)(qw >> 32) : (uint32_t)qw;
	return (int)(half >> ((p & 7) * 4)) & 15;
}

__global__ __launch_bounds__(256) void k_pack4(int n_reads, const uint8_t *seq, const int64_t *off, uint64_t *seq4, int stride)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (int64_t)n_reads * stride) return;
	const int r = (int)(t / stride), w = (int)(t % stride);
	const uint8_t *s = seq + off[r];
	const int len = (int)(off[r + 1] - off[r]);
	uint64_t v = 0;
	for (int k = 0; k < 16; ++k) {
		const int p = w * 16 + k;
		const uint64_t code = p < len ? (s[p] > 3 ? 4 : s[p]) : 15;
		v |= code << (4 * k);
	}
	seq4[t] = v;
}

// Sort one read's unsorted intervals (raw row, written by k_smem / k_smem_heavy / k_smem3) by info with the whole wavefront
// (bwamem.c:184; entries with equal info are identical, so a rank sort gives the reference's array) and write the list,
// its length and the number of SA look-ups chaining will make (bwamem.c:285-286).
__device__ __forceinline__ void wave_sort_read(const SmemLaunch &a, int rd, int lane)
{
	const int cap = a.cap, n_emit = a.raw_n[rd], n = n_emit < cap ? n_emit : cap;
	const DevIntv *Us = a.raw + (size_t)rd * cap;
	DevIntv *dst = a.out + (size_t)rd * cap;
	int n_seed = 0;
	for (int t = lane; t < n; t += 64) {
		uint64_t x0, x1, x2, info;
		get(Us + t, x0, x1, x2, info);
		int rank = 0;
		for (int u = 0; u < n; ++u) {
			uint64_t ku = get_info(Us + u);
			rank += (ku < info) || (ku == info && u < t);
		}
		put(dst + rank, x0, x1, x2, info);
		uint64_t cnt = x2;
		if (x2 > (uint64_t)a.opt.max_occ) { uint64_t step = x2 / a.opt.max_occ; cnt = (x2 + step - 1) / step; }
		n_seed += (int)(cnt < (uint64_t)a.opt.max_occ ? cnt : (uint64_t)a.opt.max_occ);
	}
	for (int m = 32; m; m >>= 1) n_seed += __shfl_xor(n_seed, m);
	if (lane == 0) {
		a.out_n[rd] = n_emit;                                // > cap tells the host to re-run with more room
		a.seed_cnt[rd] = n_emit > cap ? 0 : n_seed;
		if (n_emit > cap) atomicMax(a.worst_n, n_emit);
	}
}

// end of passes 1-2 for one read (one lane): interval count so far, diagnostics
__device__ __forceinline__ void finish_pass12(const SmemLaunch &a, int rd, int n_emit, int n_ext)
{
	a.raw_n[rd] = n_emit;
	a.l_rep[rd] = n_ext;                                     // diagnostic: bwt_extend calls passes 1-2 of this read needed
	if ((unsigned long long)n_ext > cnt_row(a.counters)[CNT_MAX_EXT]) atomicMax(&cnt_row(a.counters)[CNT_MAX_EXT], (unsigned long long)n_ext);
}

__device__ __forceinline__ int base_or_minus1(int b) { return b > 3 ? -1 : b; }

// ------------------------------------------------------------- the interval table
// The bi-interval bwt_extend (bwt.c:262) arrives at depends only on the STRING matched so far -- (first row of the string's suffix-array
// interval, first row of its reverse complement's, size) -- not on the order the bases were added in.  With 288 GB of HBM the answers for
// all strings of up to K bases are kept (K = 14: 358 M entries of 16 bytes, 5.7 GB), filled on the GPU when the index is loaded, level by
// level with the very same extension (4^L lanes, one block pair each); a string that does not occur has size 0 and so do all its extensions,
// exactly as bwt_extend returns them (only the size of such a result is ever looked at).
// Who reads it: k_smem3 (bwt_seed_strategy1 looks at no result before min_seed_len bases, so the first K - 1 extensions of every start
// are ONE look-up: 3.4 -> 1.6 ms per 1 M reads at K = 14).  k_smem does NOT: answering its short-string extensions from the table (half of
// all its calls: the first forward steps and the triangle of the backward sweep) was measured twice on the 3.1 Gbp index -- as a second
// source at the convergent step 30.4 -> 28.5 ms against 26.2 ms without the extra code (the kernel is bound by the instructions of an
// iteration, not by its Occ requests, and the iterations stay), and answered inside the run-up, one dependent gather after the other while
// the other lanes of the wavefront wait, 46 ms.
__global__ __launch_bounds__(256) void k_kmer_level(DevIndex ix, uint4 *tab, int L)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	if (L == 1) {
		const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x);
		if (c < 4) { Bi ik; set_intv(ix, c, ik); tab[kmer_off(1) + c] = pack_entry(ik.x0, ik.x1, ik.x2, 0); }
		return;
	}
	const uint64_t n_par = 1ull << 2 * (L - 1);
	const uint4 *par = tab + kmer_off(L - 1);
	uint4 *out = tab + kmer_off(L);
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_par; w += stride) {
		Bi ik; uint64_t e_;
		unpack_entry(par[w], ik.x0, ik.x1, ik.x2, e_);
		uint64_t tk[4], tl[4];
		const bool live = ik.x2 != 0;
		const uint64_t k = ik.x0 - 1, l = k + ik.x2;
		lane_occ4_pair(ix, live ? k : 0, live ? l : 0, live, tk, tl);
		const uint64_t s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
		const uint64_t b3 = ik.x1 + (ik.x0 <= ix.primary && ik.x0 + ik.x2 - 1 >= ix.primary), b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
		const uint4 zero = make_uint4(0, 0, 0, 0);
		out[w << 2 | 0] = live ? pack_entry(L2_at(ix, 0) + 1 + tk[0], b0, s0, 0) : zero;   // the string with base c put in FRONT: backward extension
		out[w << 2 | 1] = live ? pack_entry(L2_at(ix, 1) + 1 + tk[1], b1, s1, 0) : zero;
		out[w << 2 | 2] = live ? pack_entry(L2_at(ix, 2) + 1 + tk[2], b2, s2, 0) : zero;
		out[w << 2 | 3] = live ? pack_entry(L2_at(ix, 3) + 1 + tk[3], b3, s3, 0) : zero;
	}
}
// known-answer kernel: every entry of level L against the FORWARD extension of its prefix's entry (the table was filled by backward extensions:
// the two agree iff the bi-interval is a function of the string alone)
__global__ __launch_bounds__(256) void k_kmer_check(DevIndex ix, const uint4 *tab, int L, unsigned long long *bad)
{
	const uint64_t n = 1ull << 2 * L, stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n; w += stride) {
		Bi pre, got, o; uint64_t e_;
		unpack_entry(tab[kmer_off(L - 1) + (w & ((1ull << 2 * (L - 1)) - 1))], pre.x0, pre.x1, pre.x2, e_);
		unpack_entry(tab[kmer_off(L) + w], got.x0, got.x1, got.x2, e_);
		const int last = (int)(w >> 2 * (L - 1)) & 3;
		const bool live = pre.x2 != 0;
		lane_extend_c(ix, pre, 0, 3 - last, live, o);
		const bool ok = live ? (o.x2 == got.x2 && (o.x2 == 0 || (o.x0 == got.x0 && o.x1 == got.x1))) : got.x2 == 0;
		if (!ok) atomicAdd(bad, 1ull);
	}
}

// G = lanes per read: 8 (one quad per Occ block of an extend) or 4 (one quad does both blocks; 16 reads per wavefront)
template <int G>
__global__ __launch_bounds__(256, 4) void k_smem(SmemLaunch a)
{
	const int lane = lane_id();
	const int gl = lane & (G - 1);
	const int group = (int)((blockIdx.x * blockDim.x + threadIdx.x) / G);
	const DevIndex &ix = a.ix;
	const int min_seed_len = a.opt.min_seed_len, cap = a.cap, lcap = a.lcap;

	constexpr int LL = 8 * G;                                // list entries kept in LDS per read (a power of two: ring index)
	__shared__ uint4 lds_list[(256 / G) * (LL + 1)];         // +1: rows start on different banks
	uint4 *const lrow = lds_list + (threadIdx.x / G) * (LL + 1);
	uint4 *const spill = reinterpret_cast<uint4*>(a.scratch) + (size_t)group * lcap;   // list entries by physical index (lcap of them)
	DevIntv *U = a.raw;                                      // unsorted accumulated intervals of the current read (its raw row)

	// ---- per-group state (identical in every lane of the group; plain scalars so it stays in registers) ----
	int st = ST_IDLE, rd = -1, len = 0, pass = 0;
	const uint64_t *qrow = nullptr; uint64_t qw = 0; int qwi = -1;
	int x = 0, i = 0, j = 0, c = 0, min_intv = 1, ret = 0, p2k = 0, old_n = 0, out_n = 0;
	int prev_n = 0, curr_n = 0, base = 0, top = 0, mem_n = 0, mem_last_start = 0;
	uint64_t curr_last_x2 = 0, p_info = 0;
	Bi ik = { 0, 0, 0 };
	uint32_t ik_end = 0, last_push_end = 0;
	int guard = 0, guard_max = 0;
	bool exhausted = false;
	unsigned int n_ext = 0, n_blk = 0, n_out = 0;

	// Where list entry P (physical index) lives.  The backward sweep always works on the top of the region the forward pass
	// filled -- [base, top) with top fixed -- so LDS is a ring over the physical index that ends up holding the LL highest
	// entries: the forward pass writes entry P to ring slot P % LL and, from P = LL on, first moves the entry it displaces
	// (P - LL) to its HBM slot; afterwards entry P is in LDS iff P >= top - LL.  Lists of up to LL entries never touch HBM.
#define LDS_SLOT(P) ((P) & (LL - 1))
#define LIST_PUT(P, X0, X1, X2, END) do { \
		const int p_ = (P); const uint4 v_ = pack_entry((X0), (X1), (X2), (uint32_t)(END)); \
		if (p_ >= top - LL) { if (gl == 0) lrow[LDS_SLOT(p_)] = v_; } else spill[p_] = v_; } while (0)
	// (the LDS read is unconditional so that the two sources stay a ds_read_b128 and a global_load_dwordx4: written as
	// an if/else the compiler merges them into four flat_load_dword through a generic pointer)
#define LIST_GET(P, X0, X1, X2, END) do { \
		const int p_ = (P); \
		uint4 v_ = lrow[LDS_SLOT(p_)]; \
		asm volatile("" : "+v"(v_.x), "+v"(v_.y), "+v"(v_.z), "+v"(v_.w)); \
		if (p_ < top - LL) v_ = spill[p_]; \
		unpack_entry(v_, (X0), (X1), (X2), (END)); } while (0)
	// backward phase: prev[jj] is read last-pushed-first (bwt.c:321-324 reverses curr) and stays that way in place
#define PREV_AT(jj) (base + prev_n - 1 - (jj))
#define FWD_PUSH(X0, X1, X2, END) do { \
		const int p_ = curr_n; const uint4 v_ = pack_entry((X0), (X1), (X2), (uint32_t)(END)); \
		if (p_ >= LL) { uint4 ev_ = lrow[LDS_SLOT(p_)]; asm volatile("" : "+v"(ev_.x), "+v"(ev_.y), "+v"(ev_.z), "+v"(ev_.w)); spill[p_ - LL] = ev_; } \
		if (gl == 0) lrow[LDS_SLOT(p_)] = v_; \
		++curr_n; curr_last_x2 = (X2); } while (0)
#define BWD_PUSH(X0, X1, X2, END) do { LIST_PUT(PREV_AT(curr_n), (X0), (X1), (X2), (END)); ++curr_n; curr_last_x2 = (X2); } while (0)
	// kv_push(a->mem, ...) of bwamem.c:151,164,174
#define EMIT(X0, X1, X2, INFO) do { if (out_n < cap) put(U + out_n, (X0), (X1), (X2), (INFO)); ++out_n; } while (0)
	// a MEM ends at start coordinate START (bwt.c:332-336); only those >= min_seed_len are kept (bwamem.c:150,163)
#define FOUND_MEM(X0, X1, X2, FWD_END, START) do { \
		if (mem_n == 0 || (START) < mem_last_start) { \
			++mem_n; mem_last_start = (START); \
			if ((int)(uint32_t)(FWD_END) - (START) >= min_seed_len) EMIT((X0), (X1), (X2), (uint64_t)(START) << 32 | (uint32_t)(FWD_END)); \
		} } while (0)
#define QB(p) qbase(qrow, (p), qw, qwi)
#define BASE_AT(p) ((p) < 0 ? -1 : base_or_minus1(QB(p)))
	// bwt.c:289-303: start bwt_smem1a(X_, MI)
#define START_SMEM(X_, MI) do { \
		x = (X_); min_intv = (MI) < 1 ? 1 : (MI); \
		mem_n = 0; curr_n = 0; base = 0; curr_last_x2 = 0; \
		set_intv(ix, QB(x), ik); ik_end = (uint32_t)(x + 1); i = x + 1; st = ST_FWD; } while (0)
	// bwt.c:321-324: curr reversed becomes prev; its first entry is the last one pushed
#define FWD_FINISH() do { \
		ret = (int)last_push_end; \
		__builtin_amdgcn_wave_barrier(); \
		prev_n = curr_n; top = curr_n; base = 0; curr_n = 0; \
		i = x - 1; j = 0; c = BASE_AT(i); st = ST_BWD; } while (0)
	// bwt.c:343 break; pass 1 continues at the forward end (bwamem.c:146)
#define BWD_FINISH() do { st = ST_NEXT; if (pass == 1) x = ret; } while (0)

	for (;;) {
		// ---------------------------------------------------------------- fetch work
		if (st == ST_IDLE && !exhausted) {
			unsigned t = 0;
			if (gl == 0) t = atomicAdd(a.queue, 1u);
			if (G > 1) t = __shfl(t, lane & ~(G - 1));
			if (t >= (unsigned)a.n_reads) exhausted = true;
			else {
				rd = (int)t; qrow = a.seq4 + (size_t)t * a.seq4_stride; qwi = -1; U = a.raw + (size_t)t * cap; len = (int)(a.off[t + 1] - a.off[t]);
				out_n = 0; pass = 1; x = 0; guard = 0;
				guard_max = a.heavy_mult > 0 ? a.heavy_mult * len + 64 : 64 * BWAHIP_MAX_READ_LEN;
				st = len < min_seed_len ? ST_FINISH : ST_NEXT;   // bwamem.c:267
			}
		}
		if (__ballot(st != ST_IDLE) == 0) break;               // the whole wavefront is out of work

		// ---------------------------------------------------------------- run to the next bwt_extend
		bool need = false; int is_back = 0, cb = 0; Bi req = { 0, 0, 0 };
		for (int spin = 0; spin < 8192 && !need && st != ST_IDLE && st != ST_FINISH; ++spin) {
			if (st == ST_NEXT) {
				if (pass == 1) {
					while (x < len && QB(x) > 3) ++x;             // bwamem.c:145,154
					if (x < len) START_SMEM(x, 1);
					else { pass = 2; old_n = out_n < cap ? out_n : cap; p2k = 0; }
				} else if (pass == 2) {
					bool started = false;
					while (p2k < old_n && !started) {
						uint64_t x0, x1, x2, info;
						get(U + p2k, x0, x1, x2, info); ++p2k;
						int b = (int)(info >> 32), e = (int)(uint32_t)info;
						if (e - b < a.opt.split_len || x2 > (uint64_t)a.opt.split_width) continue;   // bwamem.c:160
						if (QB((b + e) >> 1) > 3) continue;       // bwt.c:296 (cannot happen inside an exact match)
						START_SMEM((b + e) >> 1, (int)x2 + 1);
						started = true;
					}
					if (!started) st = ST_FINISH;                  // pass 3 is k_smem3's
				}
			}
			// (no `else`: a start falls through to its first forward step, the end of the forward phase to the first backward one, in the same
			// turn -- the wavefront takes as many turns as its slowest lane has transitions)
			if (st == ST_FWD) {
				const int bq = i < len ? QB(i) : 4;
				if (bq < 4) { need = true; is_back = 0; req = ik; cb = 3 - bq; }
				else {                                           // end of read / ambiguous base (bwt.c:316-320)
					FWD_PUSH(ik.x0, ik.x1, ik.x2, ik_end); last_push_end = ik_end;
					FWD_FINISH();
				}
			}
			if (st == ST_BWD && !need) {
				if (c < 0) {                                     // read start or ambiguous base: every prev[] ends here
					for (int jj = 0; jj < prev_n; ++jj) {
						uint64_t x0, x1, x2, info;
						LIST_GET(PREV_AT(jj), x0, x1, x2, info);
						FOUND_MEM(x0, x1, x2, info, i + 1);
					}
					BWD_FINISH();
				} else {
					LIST_GET(PREV_AT(j), req.x0, req.x1, req.x2, p_info);
					need = true; is_back = 1; cb = c;
				}
			}
		}

		// ---------------------------------------------------------------- passes 1-2 of a read are done
		if (st == ST_FINISH) {
			if (gl == 0) { finish_pass12(a, rd, out_n, guard); n_out += out_n < cap ? out_n : cap; }
			st = ST_IDLE; rd = -1;
		}

		// ---------------------------------------------------------------- the one convergent bwt_extend
		Bi o;
		const bool live = need;
		int nb = G == 8 ? group8_extend_c(ix, req, is_back, cb, live, o) : G == 4 ? quad_extend_c(ix, req, is_back, cb, live, o) : G == 2 ? pair_extend_c(ix, req, is_back, cb, live, o)
		                                                                        : lane_extend_c(ix, req, is_back, cb, live, o);
		if (need && gl == 0) { ++n_ext; n_blk += nb; }

		// ---------------------------------------------------------------- consume the result
		if (need) {
			if (++guard > guard_max) {
				// a read deep inside a repeat: thousands of dependent steps would make it the critical path of the whole
				// launch, so it is handed to k_smem_heavy, which runs each backward step across a wavefront
				if (a.heavy_mult > 0) {
					if (gl == 0) { unsigned h = atomicAdd(a.heavy_n, 1u); a.heavy_list[h] = rd; }
					st = ST_IDLE; rd = -1;
				} else {                                             // cannot happen; guarantees the grid drains
					if (gl == 0) atomicExch(a.err, 1);
					out_n = 0; st = ST_FINISH;
				}
			} else if (st == ST_FWD) {                             // bwt.c:308-315
				bool stop = false;
				if (o.x2 != ik.x2) {
					FWD_PUSH(ik.x0, ik.x1, ik.x2, ik_end); last_push_end = ik_end;
					if (o.x2 < (uint64_t)min_intv) stop = true;
				}
				if (stop) FWD_FINISH();
				else { ik = o; ik_end = (uint32_t)(i + 1); ++i; }
			} else if (st == ST_BWD) {                             // bwt.c:328-342
				if (o.x2 < (uint64_t)min_intv) {
					if (curr_n == 0) FOUND_MEM(req.x0, req.x1, req.x2, p_info, i + 1);
				} else if (curr_n == 0 || o.x2 != curr_last_x2) {
					BWD_PUSH(o.x0, o.x1, o.x2, p_info);
				}
				if (++j == prev_n) {
					if (curr_n == 0) BWD_FINISH();
					else {
						__builtin_amdgcn_wave_barrier();
						base += prev_n - curr_n; prev_n = curr_n; curr_n = 0;   // bwt.c:340 swap, in place
						--i; j = 0; c = BASE_AT(i);
					}
				}
			}
		}
	}
	{                                                        // one set of counter updates per wavefront
		unsigned long long e = gl == 0 ? n_ext : 0, b = gl == 0 ? n_blk : 0, o = gl == 0 ? n_out : 0;
		for (int m = 32; m; m >>= 1) { e += __shfl_xor(e, m); b += __shfl_xor(b, m); o += __shfl_xor(o, m); }
		if (lane == 0 && (e | o)) {
			unsigned long long *cnt = cnt_row(a.counters);
			atomicAdd(&cnt[CNT_EXTEND], e); atomicAdd(&cnt[CNT_BLOCKS], b); atomicAdd(&cnt[CNT_INTV], o);
		}
	}
#undef PREV_AT
#undef FWD_PUSH
#undef BWD_PUSH
#undef LIST_PUT
#undef LIST_GET
#undef LDS_SLOT
#undef EMIT
#undef FOUND_MEM
#undef BASE_AT
#undef QB
#undef START_SMEM
#undef FWD_FINISH
#undef BWD_FINISH
}

// ---------------------------------------------------------------------------------------------------------------
// K1b -- the reads k_smem gave up on (more than heavy_mult x len bwt_extend calls: reads inside large repeat families).
// One read per wavefront.  The forward extension of bwt_smem1a is inherently serial and runs wavefront-uniform; the
// backward sweep (bwt.c:326-345) extends every entry of prev[] by the same base, which is independent work: lane j takes
// prev[j] and the order-dependent parts are resolved with ballots --
//   * kv_push(curr) happens iff the candidate differs in x[2] from the previous *candidate* (pushed or not: a skipped
//     candidate has the x[2] of the last pushed one), so the test is local to neighbouring candidates;
//   * a MEM is recorded only while curr is empty and only once per sweep step (the second attempt fails
//     i+1 < mem.last.start), i.e. only for j == 0.
// The list lives in LDS (in place, as in k_smem).
struct HeavyRead {
	const uint64_t *qrow; int len;
	uint4 *L; DevIntv *U;
	int out_n, ext;
	uint64_t qw; int qwi;
	unsigned n_ext, n_blk;                                   // per-lane work counters
};

__device__ __forceinline__ void heavy_emit(const SmemLaunch &a, HeavyRead &r, int lane, uint64_t x0, uint64_t x1, uint64_t x2, uint64_t info)
{
	if (r.out_n < a.cap && lane == 0) put(r.U + r.out_n, x0, x1, x2, info);
	++r.out_n;
}

// bwt_smem1a(x, min_intv) (bwt.c:285-347); returns the forward end (ret)
__device__ __forceinline__ int heavy_smem1(const SmemLaunch &a, HeavyRead &r, int lane, int x, int min_intv)
{
	const DevIndex &ix = a.ix;
	const int len = r.len, min_seed_len = a.opt.min_seed_len;
	if (min_intv < 1) min_intv = 1;
	Bi ik; set_intv(ix, qbase(r.qrow, x, r.qw, r.qwi), ik);
	uint32_t ik_end = (uint32_t)(x + 1), last_end = 0;
	int n = 0, i;
	for (i = x + 1; i < len; ++i) {                          // forward: wavefront-uniform (bwt.c:304-320)
		const int b = qbase(r.qrow, i, r.qw, r.qwi);
		if (b < 4) {
			Bi o;
			int nb = lane_extend_c(ix, ik, 0, 3 - b, true, o);
			if (lane == 0) { ++r.n_ext; r.n_blk += nb; }
			++r.ext;
			if (o.x2 != ik.x2) {
				if (lane == 0) r.L[n] = pack_entry(ik.x0, ik.x1, ik.x2, ik_end);
				++n; last_end = ik_end;
				if (o.x2 < (uint64_t)min_intv) break;
			}
			ik = o; ik_end = (uint32_t)(i + 1);
		} else {
			if (lane == 0) r.L[n] = pack_entry(ik.x0, ik.x1, ik.x2, ik_end);
			++n; last_end = ik_end;
			break;
		}
	}
	if (i == len) { if (lane == 0) r.L[n] = pack_entry(ik.x0, ik.x1, ik.x2, ik_end); ++n; last_end = ik_end; }
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

	int prev_n = n, base = 0, mem_n = 0, mem_last_start = 0;  // prev[j] = L[base + prev_n - 1 - j]
	const uint64_t lt = (1ull << lane) - 1;
	for (i = x - 1; i >= -1; --i) {                          // backward (bwt.c:326-345)
		int c = -1;
		if (i >= 0) { const int b = qbase(r.qrow, i, r.qw, r.qwi); if (b < 4) c = b; }
		uint64_t e0 = 0, e1 = 0, e2 = 0, eend = 0;
		if (c < 0) {                                         // every prev[] ends here; only prev[0] can be recorded
			unpack_entry(r.L[base + prev_n - 1], e0, e1, e2, eend);
			if (mem_n == 0 || i + 1 < mem_last_start) {
				++mem_n; mem_last_start = i + 1;
				if ((int)(uint32_t)eend - (i + 1) >= min_seed_len) heavy_emit(a, r, lane, e0, e1, e2, (uint64_t)(i + 1) << 32 | (uint32_t)eend);
			}
			break;
		}
		int curr_n = 0; bool carry_has = false; uint64_t carry_x2 = 0;
		for (int chunk = 0; chunk < prev_n; chunk += 64) {
			const int j = chunk + lane;
			const bool valid = j < prev_n;
			e0 = e1 = e2 = eend = 0;
			if (valid) unpack_entry(r.L[base + prev_n - 1 - j], e0, e1, e2, eend);
			Bi req = { e0, e1, e2 }, o;
			int nb = lane_extend_c(ix, req, 1, c, valid, o);
			if (valid) { ++r.n_ext; r.n_blk += nb; }
			r.ext += prev_n - chunk < 64 ? prev_n - chunk : 64;
			const bool isC = valid && o.x2 >= (uint64_t)min_intv;
			const uint64_t cm = __ballot(isC);
			if (chunk == 0 && !(cm & 1)) {                   // prev[0] cannot be extended and curr is still empty
				const uint64_t m0 = __shfl(e0, 0), m1 = __shfl(e1, 0), m2 = __shfl(e2, 0), mend = __shfl(eend, 0);
				if (mem_n == 0 || i + 1 < mem_last_start) {
					++mem_n; mem_last_start = i + 1;
					if ((int)(uint32_t)mend - (i + 1) >= min_seed_len) heavy_emit(a, r, lane, m0, m1, m2, (uint64_t)(i + 1) << 32 | (uint32_t)mend);
				}
			}
			const uint64_t below = cm & lt;
			const uint64_t nx2 = __shfl(o.x2, below ? 63 - __clzll((long long)below) : 0);
			const bool phas = below ? true : carry_has;
			const uint64_t px2 = below ? nx2 : carry_x2;
			const bool push = isC && (!phas || o.x2 != px2);
			const uint64_t pm = __ballot(push);
			const int pos = curr_n + __popcll(pm & lt);
			__builtin_amdgcn_wave_barrier();                 // every lane has its prev[] entry before slots are reused
			if (push) r.L[base + prev_n - 1 - pos] = pack_entry(o.x0, o.x1, o.x2, (uint32_t)eend);
			curr_n += __popcll(pm);
			if (cm) { carry_has = true; carry_x2 = __shfl(o.x2, 63 - __clzll((long long)cm)); }
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (curr_n == 0) break;
		base += prev_n - curr_n; prev_n = curr_n;            // bwt.c:340 swap, in place
	}
	return (int)last_end;
}

__global__ __launch_bounds__(256) void k_smem_heavy(SmemLaunch a)
{
	// a list per wavefront, sized for the longest read of the batch (dynamic LDS: 4 x (len + 8) x 16 bytes -- 10 KB at 150 bases, so that
	// registers, not 45 KB of LDS for the longest read the library takes, decide how many reads a CU works on)
	extern __shared__ uint4 lists_dyn[];
	const int list_cap = (a.seq4_stride - 1) * 16 + 8;
	const int lane = lane_id();
	const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * blockDim.x) >> 6);
	HeavyRead r;
	r.L = lists_dyn + (size_t)(threadIdx.x >> 6) * list_cap;
	r.n_ext = r.n_blk = 0;
	const unsigned n_heavy = *a.heavy_n;
	unsigned n_out = 0;
	// the reads come from a queue (a.queue[3]): they differ by a factor of ten in their number of extensions, and with a fixed share per
	// wavefront the kernel lasted as long as the unluckiest share
	(void)wave; (void)n_waves;
	for (;;) {
		unsigned h = 0;
		if (lane == 0) h = atomicAdd(a.queue + 3, 1u);
		h = (unsigned)__shfl((int)h, 0);
		if (h >= n_heavy) break;
		const int rd = a.heavy_list[h];
		r.qrow = a.seq4 + (size_t)rd * a.seq4_stride; r.qwi = -1; r.qw = 0;
		r.U = a.raw + (size_t)rd * a.cap;
		r.len = (int)(a.off[rd + 1] - a.off[rd]);
		r.out_n = 0; r.ext = 0;
		const int len = r.len;
		if (len >= a.opt.min_seed_len) {                     // bwamem.c:267
			for (int x = 0; x < len;) {                      // pass 1 (bwamem.c:144-154)
				if (qbase(r.qrow, x, r.qw, r.qwi) < 4) x = heavy_smem1(a, r, lane, x, 1);
				else ++x;
			}
			const int old_n = r.out_n < a.cap ? r.out_n : a.cap;   // pass 2 (bwamem.c:156-165)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			for (int k = 0; k < old_n; ++k) {
				uint64_t x0, x1, x2, info;
				get(r.U + k, x0, x1, x2, info);
				const int b = (int)(info >> 32), e = (int)(uint32_t)info;
				if (e - b < a.opt.split_len || x2 > (uint64_t)a.opt.split_width) continue;
				if (qbase(r.qrow, (b + e) >> 1, r.qw, r.qwi) > 3) continue;
				heavy_smem1(a, r, lane, (b + e) >> 1, (int)x2 + 1);
			}
		}
		if (lane == 0) finish_pass12(a, rd, r.out_n, r.ext);   // pass 3 is k_smem3's
		n_out += r.out_n < a.cap ? r.out_n : a.cap;
	}
	unsigned long long e = r.n_ext, b = r.n_blk;
	for (int m = 32; m; m >>= 1) { e += __shfl_xor(e, m); b += __shfl_xor(b, m); }
	if (lane == 0 && (e | n_out)) {
		atomicAdd(&cnt_row(a.counters)[CNT_EXTEND], e); atomicAdd(&cnt_row(a.counters)[CNT_BLOCKS], b); atomicAdd(&cnt_row(a.counters)[CNT_INTV], (unsigned long long)n_out);
		atomicAdd(&cnt_row(a.counters)[CNT_HEAVY_BLOCKS], b); atomicAdd(&cnt_row(a.counters)[CNT_HEAVY_INTV], (unsigned long long)n_out);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// K1c -- pass 3 of mem_collect_intv (bwamem.c:167-182): bwt_seed_strategy1 (bwt.c:358-380) from every position the
// previous seed ended at: forward-only extension until the interval is smaller than max_mem_intv and at least
// min_seed_len long.  Independent of passes 1-2 (it only appends to the same list), so it runs as its own lean loop:
// one read per lane, one bwt_extend per iteration, nothing but the interval and two positions as state.
__global__ __launch_bounds__(256) void k_smem3(SmemLaunch a)
{
	const DevIndex &ix = a.ix;
	const long long n_lanes = (long long)gridDim.x * blockDim.x;
	long long next = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int cap = a.cap, min_seed_len = a.opt.min_seed_len;
	const uint64_t max_intv = (uint64_t)a.opt.max_mem_intv;
	enum : int { P_IDLE = 0, P_SCAN, P_EXT };
	int st = P_IDLE, rd = 0, len = 0, x = 0, i = 0, out_n = 0;
	const uint64_t *qrow = nullptr; uint64_t qw = 0; int qwi = -1;
	DevIntv *U = a.raw;
	Bi ik = { 0, 0, 0 };
	unsigned n_ext = 0, n_blk = 0, n_new = 0, n_jump = 0;
	const int kj = ix.kmer_k < min_seed_len ? ix.kmer_k : min_seed_len;
	for (;;) {
		if (st == P_IDLE && next < a.n_reads) {
			rd = (int)next; next += n_lanes;
			len = (int)(a.off[rd + 1] - a.off[rd]);
			qrow = a.seq4 + (size_t)rd * a.seq4_stride; qwi = -1;
			U = a.raw + (size_t)rd * cap; out_n = a.raw_n[rd];
			x = 0;
			if (len >= min_seed_len) st = P_SCAN;                // bwamem.c:267: shorter reads have no intervals at all
		}
		if (__ballot(st != P_IDLE) == 0) break;
		bool need = false; int cb = 0;
		for (int spin = 0; spin < 4096 && !need && st != P_IDLE; ++spin) {
			if (st == P_SCAN) {
				while (x < len && qbase(qrow, x, qw, qwi) > 3) ++x;   // bwamem.c:170,181
				if (x >= len) { a.raw_n[rd] = out_n; st = P_IDLE; }
				else if (kj < 2) { set_intv(ix, qbase(qrow, x, qw, qwi), ik); i = x + 1; st = P_EXT; }
				else {
					// the first kj bases in one look-up of the interval table: no result before min_seed_len bases is ever looked at (bwt.c:366)
					uint32_t code = (uint32_t)qbase(qrow, x, qw, qwi);
					int m = 1, b = 4;
					while (m < kj && x + m < len && (b = qbase(qrow, x + m, qw, qwi)) < 4) { code |= (uint32_t)b << 2 * m; ++m; }
					n_jump += m - 1;
					if (m == kj) {
						uint64_t e_;
						unpack_entry(ix.kmer[kmer_off(kj) + code], ik.x0, ik.x1, ik.x2, e_);
						i = x + kj; st = P_EXT;
					} else x = x + m < len ? x + m + 1 : len;        // an ambiguous base or the read's end comes first: bwt.c:376-378
				}
			}
			if (st == P_EXT) {                                    // (no `else`: a start goes on to its first extension in the same turn)
				const int bq = i < len ? qbase(qrow, i, qw, qwi) : 4;
				if (bq < 4) { need = true; cb = 3 - bq; }
				else { x = i < len ? i + 1 : len; st = P_SCAN; }   // bwt.c:376-378
			}
		}
		Bi o;
		const bool live = need && ik.x2 != 0;                    // an empty interval stays empty: no gather needed
		const int nb = lane_extend_c(ix, ik, 0, cb, live, o);
		if (need) {
			++n_ext; n_blk += nb;
			if (o.x2 < max_intv && i - x >= min_seed_len) {       // bwt.c:366-375
				if (o.x2 > 0) {                                  // bwamem.c:174
					if (out_n < cap) put(U + out_n, o.x0, o.x1, o.x2, (uint64_t)x << 32 | (uint32_t)(i + 1));
					++out_n; ++n_new;
				}
				x = i + 1; st = P_SCAN;
			} else { ik = o; ++i; }
		}
	}
	unsigned long long e = n_ext + n_jump, b = n_blk, o = n_new, jm = n_jump;
	for (int m = 32; m; m >>= 1) { e += __shfl_xor(e, m); b += __shfl_xor(b, m); o += __shfl_xor(o, m); jm += __shfl_xor(jm, m); }
	if (lane_id() == 0 && e) {
		unsigned long long *cnt = cnt_row(a.counters);
		atomicAdd(&cnt[CNT_EXTEND], e); atomicAdd(&cnt[CNT_BLOCKS], b); atomicAdd(&cnt[CNT_INTV], o);
		if (jm) atomicAdd(&cnt[CNT_P3_JUMPED], jm);            // (the blocks of the jumped extends are not in CNT_BLOCKS: their intervals were never formed)
		atomicAdd(&cnt[CNT_P3_BLOCKS], b); atomicAdd(&cnt[CNT_P3_INTV], o);
	}
}

// K1d -- one wavefront per read: sort the raw interval row by info (bwamem.c:184) into the output row
__global__ __launch_bounds__(256) void k_intv_sort(SmemLaunch a)
{
	const int rd = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
	if (rd < a.n_reads) wave_sort_read(a, rd, lane_id());
}

} // namespace

int smem_default_groups(int G)
{
	int dev = 0, cus = 256;
	hipDeviceProp_t prop;
	if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
	// 8 waves per SIMD x 4 SIMDs = 32 waves (8 workgroups of 256) per CU, 8 groups per wave
	return cus * 16 * (64 / (G == 1 || G == 2 || G == 4 ? G : 8));   // 4 waves per SIMD are resident (VGPRs)
}

int launch_pack4(const SmemLaunch &a, hipStream_t st)
{
	const int64_t words = (int64_t)a.n_reads * a.seq4_stride;
	hipLaunchKernelGGL(k_pack4, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, a.n_reads, a.seq, a.off, a.seq4, a.seq4_stride);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_kmer_table(const DevIndex &ix, uint4 *tab, int K, hipStream_t st)
{
	for (int L = 1; L <= K; ++L) {
		const uint64_t n_par = L == 1 ? 4 : 1ull << 2 * (L - 1);
		const unsigned blocks = (unsigned)std::min<uint64_t>((n_par + 255) / 256, 256 * 16);
		hipLaunchKernelGGL(k_kmer_level, dim3(blocks), dim3(256), 0, st, ix, tab, L);
	}
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
int launch_kmer_check(const DevIndex &ix, int L, unsigned long long *bad, hipStream_t st)
{
	if (L < 2 || L > ix.kmer_k) return BWAHIP_EINVAL;
	const unsigned blocks = (unsigned)std::min<uint64_t>(((1ull << 2 * L) + 255) / 256, 256 * 16);
	hipLaunchKernelGGL(k_kmer_check, dim3(blocks), dim3(256), 0, st, ix, ix.kmer, L, bad);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_smem3(const SmemLaunch &a, hipStream_t st)
{
	if (a.opt.max_mem_intv > 0) {
		int blocks = (a.n_reads + 255) / 256;
		if (blocks > 256 * 8) blocks = 256 * 8;                  // 8 waves per SIMD resident; lanes stride over the reads
		hipLaunchKernelGGL(k_smem3, dim3(blocks), dim3(256), 0, st, a);
	}
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_intv_sort(const SmemLaunch &a, hipStream_t st)
{
	hipLaunchKernelGGL(k_intv_sort, dim3((a.n_reads + 3) / 4), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_smem_heavy(const SmemLaunch &a, hipStream_t st)
{
	int waves = a.groups_total < 5120 ? a.groups_total : 5120;   // one scratch region per wavefront; 5 wavefronts per SIMD fit (99 VGPRs)
	if (waves < 4) return BWAHIP_EINVAL;
	const size_t lds = (size_t)4 * ((a.seq4_stride - 1) * 16 + 8) * 16;
	hipLaunchKernelGGL(k_smem_heavy, dim3(waves / 4), dim3(256), lds, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_smem(const SmemLaunch &a, int G, hipStream_t st)
{
	int groups = a.groups_total;
	int blocks = groups / (256 / (G == 1 || G == 2 || G == 4 ? G : 8));
	if (blocks < 1) blocks = 1;
	if (G == 1) hipLaunchKernelGGL(k_smem<1>, dim3(blocks), dim3(256), 0, st, a);
	else if (G == 2) hipLaunchKernelGGL(k_smem<2>, dim3(blocks), dim3(256), 0, st, a);
	else if (G == 4) hipLaunchKernelGGL(k_smem<4>, dim3(blocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL(k_smem<8>, dim3(blocks), dim3(256), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
