/* ORACLE (test infrastructure only) -- prototypes.  See ora_types.h header. */
#ifndef ORA_H
#define ORA_H
#include "ora_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ora_index.c -- stock `bwa index` file formats (bwt.c:385-462, bntseq.c:65-211, bwa.c:402-434) */
ora_index_t *ora_index_load(const char *prefix);
void ora_index_destroy(ora_index_t *idx);
extern const uint8_t ora_nt4_table[256];                 /* bntseq.c:46 */
int ora_pos2rid(const ora_ref_t *r, int64_t pos_f);      /* bntseq.c:354 */
int ora_intv2rid(const ora_ref_t *r, int64_t rb, int64_t re);   /* bntseq.c:370 */
uint8_t *ora_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len);   /* bntseq.c:403 */
uint8_t *ora_fetch_seq(const ora_ref_t *r, int64_t *beg, int64_t mid, int64_t *end, int *rid);      /* bntseq.c:426 */
static inline int64_t ora_depos(const ora_ref_t *r, int64_t pos, int *is_rev)                       /* bntseq.h:87 */
{
	return (*is_rev = (pos >= r->l_pac)) ? (r->l_pac << 1) - 1 - pos : pos;
}

/* ora_fmi.c */
void ora_occ4(const ora_fmi_t *f, uint64_t k, uint64_t cnt[4]);
uint64_t ora_occ(const ora_fmi_t *f, uint64_t k, int c);
void ora_set_intv(const ora_fmi_t *f, int c, ora_intv_t *ik);
void ora_extend(const ora_fmi_t *f, const ora_intv_t *ik, ora_intv_t ok[4], int is_back);
int ora_smem1a(const ora_fmi_t *f, int len, const uint8_t *q, int x, int min_intv, uint64_t max_intv, ora_intv_v *mem, ora_intv_v *tmp[2]);
int ora_seed_strategy1(const ora_fmi_t *f, int len, const uint8_t *q, int x, int min_len, int max_intv, ora_intv_t *mem);
uint64_t ora_sa(const ora_fmi_t *f, uint64_t k);

/* ora_ksw.c */
int ora_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
                    int *qle, int *tle, int *gtle, int *gscore, int *max_off);                       /* ksw.c:380 */
int ora_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar, uint32_t **cigar); /* ksw.c:504 */
ora_kswr_t ora_ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat,
                          int o_del, int e_del, int o_ins, int e_ins, int xtra);                      /* ksw.c:343 */

/* ora_chain.c */
typedef struct { ora_intv_v mem, mem1, tmpv[2]; } ora_aux_t;                                          /* bwa.h:205 */
ora_aux_t *ora_aux_new(void);
void ora_aux_free(ora_aux_t *a);
void ora_opt_init(ora_opt_t *o);                                                                      /* bwamem.c:74 */
void ora_fill_scmat(int a, int b, int8_t mat[25]);                                                    /* bwa.c:249 */
void ora_collect_intv(const ora_opt_t *opt, const ora_fmi_t *f, int len, const uint8_t *seq, ora_aux_t *a);   /* bwamem.c:137 */
ora_chain_v ora_chain(const ora_opt_t *opt, const ora_index_t *idx, int len, const uint8_t *seq, ora_aux_t *aux); /* bwamem.c:258 */
int ora_chain_weight(const ora_chain_t *c);                                                           /* bwamem.c:220 */
int ora_chain_flt(const ora_opt_t *opt, int n_chn, ora_chain_t *a);                                   /* bwamem.c:334 */
void ora_flt_chained_seeds(const ora_opt_t *opt, const ora_ref_t *r, int l_query, const uint8_t *query, int n_chn, ora_chain_t *a); /* bwamem.c:605 */

/* ora_extend.c */
void ora_chain2aln(const ora_opt_t *opt, const ora_ref_t *r, int l_query, const uint8_t *query, const ora_chain_t *c, ora_reg_v *av); /* bwamem.c:639 */
int ora_sort_dedup_patch(const ora_opt_t *opt, const ora_ref_t *r, uint8_t *query, int n, ora_reg_t *a);                          /* bwamem.c:444 */
uint32_t *ora_gen_cigar2(const int8_t mat[25], int o_del, int e_del, int o_ins, int e_ins, int w_, int64_t l_pac, const uint8_t *pac,
                         int l_query, uint8_t *query, int64_t rb, int64_t re, int *score, int *n_cigar, int *NM);                 /* bwa.c:261 */
ora_reg_v ora_align1_core(const ora_opt_t *opt, const ora_index_t *idx, int l_seq, char *seq, ora_aux_t *aux);                    /* bwamem.c:1061 */

/* ora_final.c */
int ora_mark_primary_se(const ora_opt_t *opt, int n, ora_reg_t *a, int64_t id);                        /* bwamem.c:528 */
int ora_approx_mapq_se(const ora_opt_t *opt, const ora_reg_t *a);                                      /* bwamem.c:962 */
void ora_reorder_primary5(int T, ora_reg_v *a);                                                        /* bwamem.c:988 */
ora_aln_t ora_reg2aln(const ora_opt_t *opt, const ora_ref_t *r, int l_query, const char *query, const ora_reg_t *ar); /* bwamem.c:1099 */
void ora_aln2sam(const ora_opt_t *opt, const ora_ref_t *r, ora_str_t *str, ora_read_t *s, int n, const ora_aln_t *list, int which, const ora_aln_t *m); /* bwamem.c:832 */
void ora_reg2sam(const ora_opt_t *opt, const ora_ref_t *r, ora_read_t *s, ora_reg_v *a, int extra_flag, const ora_aln_t *m); /* bwamem.c:1013 */
char **ora_gen_alt(const ora_opt_t *opt, const ora_ref_t *r, const ora_reg_v *a, int l_query, const char *query); /* bwamem_extra.c:124 */
extern char ora_rg_id[256];                                                                            /* bwa.c:44 */

/* ora_pair.c */
void ora_pestat(const ora_opt_t *opt, int64_t l_pac, int n, const ora_reg_v *regs, ora_pestat_t pes[4]);      /* bwamem_pair.c:72 */
int ora_matesw(const ora_opt_t *opt, const ora_ref_t *r, const ora_pestat_t pes[4], const ora_reg_t *a, int l_ms, const uint8_t *ms, ora_reg_v *ma); /* bwamem_pair.c:137 */
int ora_pair(const ora_opt_t *opt, const ora_ref_t *r, const ora_pestat_t pes[4], ora_read_t s[2], ora_reg_v a[2], int id, int *sub, int *n_sub, int z[2], int n_pri[2]); /* bwamem_pair.c:208 */
int ora_sam_pe(const ora_opt_t *opt, const ora_ref_t *r, const ora_pestat_t pes[4], uint64_t id, ora_read_t s[2], ora_reg_v a[2]); /* bwamem_pair.c:276 */

/* ora_process.c */
void ora_process_seqs(const ora_opt_t *opt, const ora_index_t *idx, int64_t n_processed, int n, ora_read_t *seqs, const ora_pestat_t *pes0); /* bwamem.c:1215 */

/* helpers shared by the files above */
void ora_str_putc(ora_str_t *s, int c);
void ora_str_putsn(ora_str_t *s, const char *p, int l);
void ora_str_puts(ora_str_t *s, const char *p);
void ora_str_putw(ora_str_t *s, int v);       /* kstring.h kputw */
void ora_str_putl(ora_str_t *s, long v);      /* kstring.h kputl */
static inline uint64_t ora_hash64(uint64_t key)                                                        /* utils.h:97 */
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}
void ora_sort_u64(size_t n, uint64_t *a);     /* utils.c:47 ks_introsort_64 */
typedef struct { uint64_t x, y; } ora_pair64_t;
void ora_sort_pair64(size_t n, ora_pair64_t *a); /* utils.c:46 ks_introsort_128 */

#ifdef __cplusplus
}
#endif
#endif
