/* bwamem_hip.h -- libbwamem_hip.so: the reference's own entry point, same name and signature, MI355X underneath.
 *
 *     void mem_process_seqs(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac,
 *                           int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0);
 *                                                            -- declaration bwamem.h:69, definition bwamem.c:1215
 *
 * A reference translation unit keeps including its own bwamem.h (the prototype there is this very function) and links
 * libbwamem_hip.so + libbwahip.so in place of bwamem.c's definition; INTEGRATION.md section 1 shows the two-line
 * Makefile change.  This header is for callers that do not have the reference headers: it declares the same function
 * over the layout mirrors of include/bwahip.h.
 */
#ifndef BWAMEM_HIP_H
#define BWAMEM_HIP_H
#include "bwahip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* mem_process_seqs (bwamem.h:69): aligns seqs[0..n) (interleaved mates when opt->flag & MEM_F_PE), sets seqs[i].sam to
 * malloc()ed SAM text, overwrites seqs[i].seq with 0..4 codes.  No return value; failures are fatal like the reference's
 * err_fatal (utils.c:90-99).  The index goes to the GPU on the first call and is reused while the same pointers come in. */
void mem_process_seqs(const bwahip_opt_t *opt, const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac,
                      int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0);

/* bwa_rg_id (bwa.c:44) is a global of the reference that mem_aln2sam reads (bwamem.c:920); forward -R's id here. */
void bwahip_compat_set_rg_id(const char *id);
/* Destroy the cached GPU context (also done at exit). */
void bwahip_compat_release(void);

#ifdef __cplusplus
}
#endif
#endif
