// K9 -- SAM text of single-end reads on the GPU: mem_reg2sam's record list (bwamem.c:1025-1056: supplementary flag, mapQ cap
// of supplementary hits, XS of secondaries, XA) and mem_aln2sam (sam_dev.h).  One read per wavefront.  Two launches of the
// same code: the first computes each read's text length (and applies the per-record adjustments once), an exclusive
// scan turns lengths into offsets, the second writes -- so the text of read i always lands at a fixed place and the
// batch's SAM is one contiguous buffer in read order (no atomics: deterministic output).
#include "bwahip_internal.h"
#include "sam_dev.h"

namespace {
using namespace samdev;

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }

template <bool WRITE>
__global__ __launch_bounds__(64) void k_sam_se(FinLaunch a)
{
	const int r = (int)blockIdx.x + a.read_lo, l = lane();
	const DevOpt &opt = a.opt;
	const int n = a.freg_n[r];
	const int64_t rb0 = a.reg_base[r];
	const FinReg *f = a.fregs + rb0;
	const uint8_t *need = a.need + rb0;
	const int *aln_of = a.aln_of_reg + rb0;
	const DevAln **list = a.rec_list + rb0;                     // this read's records / XA members: pointer lists in its region slots
	const DevAln **xa = a.xa_list + rb0;
	int n_rec = 0;
	if (!WRITE) {
		// mem_reg2sam's adjustments of the record copies (bwamem.c:1033-1041), applied once to the alignment array
		if (l == 0) {
			int mapq0 = 0;
			for (int k = 0; k < n; ++k) {
				if (!(need[k] & NEED_REC)) continue;
				DevAln *q = a.alns + aln_of[k];
				if (f[k].secondary >= 0) q->sub = -1;
				if (n_rec && f[k].secondary < 0) q->flag |= (opt.flag & BWAHIP_F_NO_MULTI) ? 0x10000 : 0x800;
				if (!(opt.flag & BWAHIP_F_KEEP_SUPP_MAPQ) && n_rec && !f[k].is_alt && q->mapq > (uint32_t)mapq0) q->mapq = (uint32_t)mapq0;
				if (n_rec == 0) mapq0 = (int)q->mapq;
				list[n_rec++] = q;
			}
		}
		n_rec = __shfl(n_rec, 0);
		__threadfence_block(); __syncthreads();
	} else {
		n_rec = a.rec_n[r];                                       // the list was built by the sizing pass
	}
	Tables t = { a.ctg_names, a.ctg_name_off, a.ctg_anno, a.ctg_anno_off, a.pool, a.rg_id, a.rg_len, opt.flag };
	ReadText s;
	s.name = a.names + a.name_off[r];
	s.comment = a.comments && a.comment_off[r + 1] > a.comment_off[r] ? a.comments + a.comment_off[r] : nullptr;
	s.seq = a.seq + a.off[r]; s.qual = a.qual && a.qual_off[r] >= 0 ? a.qual + a.qual_off[r] : nullptr; s.l_seq = (int)(a.off[r + 1] - a.off[r]);
	Emit e = { WRITE ? a.sam + a.sam_off[r] : nullptr, 0, l };
	if (n_rec == 0) {
		// unaligned read (bwamem.c:1043-1047): mem_reg2aln(..., 0) gives rid = pos = -1, flag 0x4, everything else 0
		__shared__ __attribute__((aligned(16))) DevAln s_un;
		__shared__ const DevAln *s_unp;
		if (l == 0) {
			DevAln u;
			memset(&u, 0, sizeof u);
			u.rid = -1; u.pos = -1; u.flag = 0x4;
			s_un = u; s_unp = &s_un;
		}
		__syncthreads();
		emit_record(e, t, s, 1, &s_unp, 0, nullptr, 0, nullptr);
	} else {
		int which = 0;
		for (int k = 0; k < n; ++k) {
			if (!(need[k] & NEED_REC)) continue;
			// XA members of record k: regions i (ascending) whose owner is k (bwamem_extra.c:141-160)
			int n_xa = 0;
			if (!(opt.flag & BWAHIP_F_ALL)) {
				if (l == 0) for (int i = 0; i < n; ++i) if ((need[i] & NEED_XA) && a.xa_owner[rb0 + i] == k) xa[n_xa++] = a.alns + aln_of[i];
				n_xa = __shfl(n_xa, 0);
				__threadfence_block(); __syncthreads();
			}
			emit_record(e, t, s, n_rec, list, which, nullptr, n_xa, xa);
			++which;
			__syncthreads();
		}
	}
	if (!WRITE && l == 0) { a.sam_len[r] = (int)e.pos; a.rec_n[r] = n_rec; }
}


__device__ __forceinline__ int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)   // bwamem_pair.c:48
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// SAM text of one end of a pair (mem_sam_pe's output part, bwamem_pair.c:366-385 and 397-418).  One read per wavefront; the
// decisions come from k_pair (PeRead), the mate's best hit h[!i] is attached to every record (mate fields, MC, TLEN).
template <bool WRITE>
__global__ __launch_bounds__(64, (WRITE ? 6 : 8)) void k_sam_pe(FinLaunch a)   // (occupancy over registers: size pass at 8 waves per SIMD 1.56 -> 1.30 ms, write pass at 6 with 48 B of spill 3.47 -> 3.27 ms)
{
	__shared__ __attribute__((aligned(16))) DevAln s_un[2];      // [0] unaligned record of this read, [1] unaligned mate (DevAln is padded to 80 bytes)
	__shared__ const DevAln *s_unp;
	const int r = (int)blockIdx.x + a.read_lo, l = lane(), rm = r ^ 1, end = r & 1;
	const DevOpt &opt = a.opt;
	const int n = a.freg_n[r];
	const int64_t rb0 = a.reg_base[r], rbm = a.reg_base[rm];
	const FinReg *f = a.fregs + rb0;
	const uint8_t *need = a.need + rb0;
	const int *aln_of = a.aln_of_reg + rb0;
	const DevAln **list = a.rec_list + rb0;
	const DevAln **xa = a.xa_list + rb0;
	const PeRead pr = a.pe_read[r], prm = a.pe_read[rm];
	if (l == 0) {
		DevAln u;
		memset(&u, 0, sizeof u);
		u.rid = -1; u.pos = -1; u.flag = 0x4;
		s_un[0] = u; s_un[1] = u; s_unp = &s_un[0];
	}
	__syncthreads();
	const DevAln *m = prm.h_reg >= 0 ? a.alns + a.aln_of_reg[rbm + prm.h_reg] : &s_un[1];
	const DevAln *h = pr.h_reg >= 0 ? a.alns + aln_of[pr.h_reg] : &s_un[0];
	int extra = pr.extra_flag;
	if (pr.mode == 0) {
		// proper-pair bit of the unpaired path (bwamem_pair.c:406-411): the two best hits on one contig within the insert range
		if (!(opt.flag & BWAHIP_F_NOPAIRING) && h->rid == m->rid && h->rid >= 0) {
			int64_t dist;
			const int64_t b_own = f[0].rb, b_mate = a.fregs[rbm].rb;
			const int d = end == 0 ? infer_dir(a.ix.l_pac, b_own, b_mate, &dist) : infer_dir(a.ix.l_pac, b_mate, b_own, &dist);
			if (!a.pes[d].failed && dist >= a.pes[d].low && dist <= a.pes[d].high) extra |= 2;
		}
		extra |= end == 0 ? 0x40 : 0x80;
	}
	int n_rec = 0;
	if (!WRITE) {
		if (l == 0) {
			if (pr.mode == 1) {                                   // bwamem_pair.c:366-377: h[i], then the ALT hit as supplementary
				DevAln *q = a.alns + aln_of[pr.h_reg];
				q->mapq = (uint32_t)pr.mapq & 0xff;
				q->flag |= (0x40 << end) | extra;
				list[n_rec++] = q;
				if (pr.alt_reg >= 0) {
					DevAln *g = a.alns + aln_of[pr.alt_reg];
					g->flag |= 0x800 | (0x40 << end) | extra;
					list[n_rec++] = g;
				}
			} else {                                              // mem_reg2sam with extra_flag and the mate (bwamem.c:1033-1041)
				int mapq0 = 0;
				for (int k = 0; k < n; ++k) {
					if (!(need[k] & NEED_REC)) continue;
					DevAln *q = a.alns + aln_of[k];
					q->flag |= extra;
					if (f[k].secondary >= 0) q->sub = -1;
					if (n_rec && f[k].secondary < 0) q->flag |= (opt.flag & BWAHIP_F_NO_MULTI) ? 0x10000 : 0x800;
					if (!(opt.flag & BWAHIP_F_KEEP_SUPP_MAPQ) && n_rec && !f[k].is_alt && q->mapq > (uint32_t)mapq0) q->mapq = (uint32_t)mapq0;
					if (n_rec == 0) mapq0 = (int)q->mapq;
					list[n_rec++] = q;
				}
			}
		}
		n_rec = __shfl(n_rec, 0);
		__threadfence_block(); __syncthreads();
	} else n_rec = a.rec_n[r];
	Tables t = { a.ctg_names, a.ctg_name_off, a.ctg_anno, a.ctg_anno_off, a.pool, a.rg_id, a.rg_len, opt.flag };
	ReadText s;
	s.name = a.names + a.name_off[r];
	s.comment = a.comments && a.comment_off[r + 1] > a.comment_off[r] ? a.comments + a.comment_off[r] : nullptr;
	s.seq = a.seq + a.off[r]; s.qual = a.qual && a.qual_off[r] >= 0 ? a.qual + a.qual_off[r] : nullptr; s.l_seq = (int)(a.off[r + 1] - a.off[r]);
	Emit e = { WRITE ? a.sam + a.sam_off[r] : nullptr, 0, l };
	if (n_rec == 0) {
		if (l == 0) s_un[0].flag = 0x4 | extra;                     // t.flag |= extra_flag (bwamem.c:1045)
		__syncthreads();
		emit_record(e, t, s, 1, &s_unp, 0, m, 0, nullptr);
	} else {
		for (int which = 0; which < n_rec; ++which) {
			// the region this record came from: its XA members are the regions it owns
			int k_reg = -1, n_xa = 0;
			if (pr.mode == 1) k_reg = which == 0 ? pr.h_reg : pr.alt_reg;
			else { int c = 0; for (int k = 0; k < n; ++k) if (need[k] & NEED_REC) { if (c == which) { k_reg = k; break; } ++c; } }
			if (!(opt.flag & BWAHIP_F_ALL)) {
				if (l == 0) for (int i = 0; i < n; ++i) if ((need[i] & NEED_XA) && a.xa_owner[rb0 + i] == k_reg) xa[n_xa++] = a.alns + aln_of[i];
				n_xa = __shfl(n_xa, 0);
				__threadfence_block(); __syncthreads();
			}
			emit_record(e, t, s, n_rec, list, which, m, n_xa, xa);
			__syncthreads();
		}
	}
	if (!WRITE && l == 0) { a.sam_len[r] = (int)e.pos; a.rec_n[r] = n_rec; }
}

} // namespace

int launch_sam_pe(const FinLaunch &a_, bool write, hipStream_t st, int read_lo, int read_hi)
{
	if (read_hi < 0) read_hi = a_.n_reads;
	if (read_hi <= read_lo) return 0;
	FinLaunch a = a_;
	a.read_lo = read_lo;
	if (write) hipLaunchKernelGGL(k_sam_pe<true>, dim3(read_hi - read_lo), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_sam_pe<false>, dim3(read_hi - read_lo), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}

int launch_sam(const FinLaunch &a_, bool write, hipStream_t st, int read_lo, int read_hi)
{
	if (read_hi < 0) read_hi = a_.n_reads;
	if (read_hi <= read_lo) return 0;
	FinLaunch a = a_;
	a.read_lo = read_lo;
	if (write) hipLaunchKernelGGL(k_sam_se<true>, dim3(read_hi - read_lo), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_sam_se<false>, dim3(read_hi - read_lo), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
