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
	const int r = blockIdx.x, l = lane();
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
		__shared__ DevAln s_un;
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

} // namespace

int launch_sam(const FinLaunch &a, bool write, hipStream_t st)
{
	if (a.n_reads <= 0) return 0;
	if (write) hipLaunchKernelGGL(k_sam_se<true>, dim3(a.n_reads), dim3(64), 0, st, a);
	else hipLaunchKernelGGL(k_sam_se<false>, dim3(a.n_reads), dim3(64), 0, st, a);
	return hipGetLastError() == hipSuccess ? 0 : BWAHIP_ENODEV;
}
