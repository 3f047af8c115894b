/* ORACLE -- TEST INFRASTRUCTURE ONLY.  A plain-C CPU restatement of the
 * reference's bwa-mem CPU algorithm (bwa 0.7.17 as found in /root/reference).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link or execute anything in oracle/.  The product (bwa-mem-gpu_amd/)
 * never includes or links these files.
 *
 * Parity status: PINNED against the reference itself (oracle/_ref/bwaref, built
 * from the reference's own sources) by tests/test_oracle_vs_ref.py and by the
 * committed golden vectors in tests/golden/ that bwaref generated.
 */
#ifndef ORA_TYPES_H
#define ORA_TYPES_H

#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

/* ---- FM index (reference: bwt.h:48-64) ---- */
typedef struct {
	uint64_t primary;    /* row of '$' */
	uint64_t L2[5];      /* cumulative base counts */
	uint64_t seq_len;    /* 2 * l_pac */
	uint64_t n_words;    /* uint32 words in occ-interleaved bwt */
	uint32_t *bwt;       /* per 128 bases: 4 x u64 counts + 8 x u32 packed bases (bwtindex.c:150) */
	int sa_intv;
	uint64_t n_sa;
	uint64_t *sa;        /* sa[0] = (uint64_t)-1 (bwt.c:83) */
} ora_fmi_t;

typedef struct { uint64_t x[3], info; } ora_intv_t;     /* bwt.h:62 */
typedef struct { int n, m; ora_intv_t *a; } ora_intv_v;

/* ---- reference sequence annotations (bntseq.h:41-64) ---- */
typedef struct {
	int64_t offset;
	int32_t len, n_ambs;
	uint32_t gi;
	int32_t is_alt;
	char *name, *anno;
} ora_ann_t;

typedef struct { int64_t offset; int32_t len; char amb; } ora_amb_t;

typedef struct {
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	ora_ann_t *anns;
	int32_t n_holes;
	ora_amb_t *ambs;
	uint8_t *pac;        /* forward strand, 4 bases / byte, MSB first (bntseq.c:229) */
} ora_ref_t;

typedef struct { ora_fmi_t *fmi; ora_ref_t *ref; } ora_index_t;

/* ---- options (bwa.h:86-118; defaults bwamem.c:74-110) ---- */
#define ORA_F_PE        0x2
#define ORA_F_NOPAIRING 0x4
#define ORA_F_ALL       0x8
#define ORA_F_NO_MULTI  0x10
#define ORA_F_NO_RESCUE 0x20
#define ORA_F_REF_HDR   0x100
#define ORA_F_SOFTCLIP  0x200
#define ORA_F_SMARTPE   0x400
#define ORA_F_PRIMARY5  0x800
#define ORA_F_KEEP_SUPP_MAPQ 0x1000
#define ORA_F_XB        0x2000

typedef struct {
	uint64_t max_mem_intv;
	int a, b, o_del, e_del, o_ins, e_ins;
	int pen_unpaired, pen_clip5, pen_clip3;
	int w, zdrop, T, flag;
	int min_seed_len, min_chain_weight, max_chain_extend;
	float split_factor;
	int split_width, max_occ, max_chain_gap, n_threads, chunk_size;
	float mask_level, drop_ratio, XA_drop_ratio, mask_level_redun, mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins, max_matesw, max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} ora_opt_t;

/* ---- chaining (bwa.h:121-142) ---- */
typedef struct { int64_t rbeg; int32_t qbeg, len; int score; } ora_seed_t;

typedef struct {
	int n, m;
	ora_seed_t *seeds;
	int64_t pos;
	int first, rid;
	uint32_t w;          /* 29-bit in the reference */
	int kept, is_alt;
	float frac_rep;
} ora_chain_t;
typedef struct { int n, m; ora_chain_t *a; } ora_chain_v;

/* ---- alignment regions (bwa.h:145-165) ---- */
typedef struct {
	int64_t rb, re;
	uint64_t hash;
	float frac_rep;
	int qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov;
	int secondary, secondary_all, seedlen0;
	int n_comp, is_alt;
} ora_reg_t;
typedef struct { int n, m; ora_reg_t *a; } ora_reg_v;

typedef struct { int low, high, failed; double avg, std; } ora_pestat_t;  /* bwa.h:167-171 */

typedef struct {                 /* bwa.h:173-184 */
	int64_t pos;
	char *XA;
	uint32_t *cigar;             /* followed in the same buffer by the MD string (bwa.c:311) */
	int rid, flag;
	uint32_t is_rev, is_alt, mapq, NM;
	int n_cigar;
	int score, sub, alt_sc;
} ora_aln_t;

typedef struct {                 /* bwa.h:58-63 */
	int l_seq, id;
	char *name, *comment, *seq, *qual, *sam;
} ora_read_t;

/* growable byte string (kstring.h) */
typedef struct { size_t l, m; char *s; } ora_str_t;

/* SSE2-exact local alignment result (ksw.h:42-48) */
typedef struct { int score, te, qe, score2, te2, tb, qb; } ora_kswr_t;
#define ORA_KSW_XBYTE  0x10000
#define ORA_KSW_XSTOP  0x20000
#define ORA_KSW_XSUBO  0x40000
#define ORA_KSW_XSTART 0x80000

#endif
