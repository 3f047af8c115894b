/* TEST INFRASTRUCTURE ONLY.
 *
 * Command-line options of `bwa mem` for the two checker front ends (main_oracle.c over ora_opt_t, ref_driver.c over
 * the reference's own mem_opt_t): same letters, same meaning and the same "scale by -A unless set" rule as the
 * reference's main_mem (fastmap.c:43-57 update_a, fastmap.c:77-175 getopt loop, fastmap.c:240-269 -x presets), so that
 * a parity test can hand both binaries one option string.  The two option structs have identical field names; the
 * includer defines OPT_T / PES_T (struct types), OPT_FILL_SCMAT(a, b, mat) and OPT_LOG(x) before including this file.
 *
 * Flags that only matter to I/O of the real CLI (-o -f -H -v -1) are not accepted.
 */
#ifndef ORA_OPT_PARSE_H
#define ORA_OPT_PARSE_H
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <unistd.h>

#define OPT_GETOPT_STRING "Z5qpaMCSPVYjuk:c:s:r:t:R:A:B:O:E:U:w:L:d:T:Q:D:m:I:N:W:x:G:h:y:K:X:"

typedef struct {
	OPT_T *opt;
	OPT_T set;               /* non-zero field = given on the command line (the reference's opt0) */
	const char *mode;        /* -x */
	int fixed_chunk;         /* -K */
	int copy_comment;        /* -C */
	int ignore_alt;          /* -j */
	int smart_pe;            /* -p */
	int has_pes0;            /* -I */
	int align_only;          /* -Z (ours): time the per-read hot path only (worker1 / mem_align1_core), no finalisation, no SAM */
	PES_T pes[4];
	char rg_id[256];         /* ID: field of -R (bwa_set_rg, bwa.c:562-583) */
} optparse_t;

/* "INT[,INT]": second value only when a punctuation mark is followed by a digit (fastmap.c:131-150) */
static void opt_pair(const char *arg, int *first, int *second)
{
	char *p;
	*first = *second = (int)strtol(arg, &p, 10);
	if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) *second = (int)strtol(p + 1, &p, 10);
}

static void optparse_init(optparse_t *x, OPT_T *opt)
{
	int i;
	memset(x, 0, sizeof *x);
	x->opt = opt;
	for (i = 0; i < 4; ++i) x->pes[i].failed = 1;
}

/* one getopt result; returns 0 if the letter was taken */
static int optparse_one(optparse_t *x, int c, const char *arg)
{
	OPT_T *o = x->opt, *s = &x->set;
	switch (c) {
	case 'k': o->min_seed_len = atoi(arg); s->min_seed_len = 1; break;
	case 'w': o->w = atoi(arg); s->w = 1; break;
	case 'A': o->a = atoi(arg); s->a = 1; break;
	case 'B': o->b = atoi(arg); s->b = 1; break;
	case 'T': o->T = atoi(arg); s->T = 1; break;
	case 'U': o->pen_unpaired = atoi(arg); s->pen_unpaired = 1; break;
	case 't': o->n_threads = atoi(arg); if (o->n_threads < 1) o->n_threads = 1; break;
	case 'P': o->flag |= 0x4; break;                       /* MEM_F_NOPAIRING */
	case 'a': o->flag |= 0x8; break;                       /* MEM_F_ALL */
	case 'p': o->flag |= 0x2 | 0x400; x->smart_pe = 1; break;   /* MEM_F_PE | MEM_F_SMARTPE */
	case 'M': o->flag |= 0x10; break;                      /* MEM_F_NO_MULTI */
	case 'S': o->flag |= 0x20; break;                      /* MEM_F_NO_RESCUE */
	case 'Y': o->flag |= 0x200; break;                     /* MEM_F_SOFTCLIP */
	case 'V': o->flag |= 0x100; break;                     /* MEM_F_REF_HDR */
	case '5': o->flag |= 0x800 | 0x1000; break;            /* MEM_F_PRIMARY5 | MEM_F_KEEP_SUPP_MAPQ */
	case 'q': o->flag |= 0x1000; break;                    /* MEM_F_KEEP_SUPP_MAPQ */
	case 'u': o->flag |= 0x2000; break;                    /* MEM_F_XB */
	case 'c': o->max_occ = atoi(arg); s->max_occ = 1; break;
	case 'd': o->zdrop = atoi(arg); s->zdrop = 1; break;
	case 'j': x->ignore_alt = 1; break;
	case 'r': o->split_factor = (float)atof(arg); s->split_factor = 1.f; break;
	case 'D': o->drop_ratio = (float)atof(arg); s->drop_ratio = 1.f; break;
	case 'm': o->max_matesw = atoi(arg); s->max_matesw = 1; break;
	case 's': o->split_width = atoi(arg); s->split_width = 1; break;
	case 'G': o->max_chain_gap = atoi(arg); s->max_chain_gap = 1; break;
	case 'N': o->max_chain_extend = atoi(arg); s->max_chain_extend = 1; break;
	case 'W': o->min_chain_weight = atoi(arg); s->min_chain_weight = 1; break;
	case 'y': o->max_mem_intv = (uint64_t)atol(arg); s->max_mem_intv = 1; break;
	case 'C': x->copy_comment = 1; break;
	case 'Z': x->align_only = 1; break;
	case 'K': x->fixed_chunk = atoi(arg); break;
	case 'X': o->mask_level = (float)atof(arg); break;
	case 'x': x->mode = arg; break;
	case 'h': opt_pair(arg, &o->max_XA_hits, &o->max_XA_hits_alt); s->max_XA_hits = s->max_XA_hits_alt = 1; break;
	case 'Q':
		s->mapQ_coef_len = 1;
		o->mapQ_coef_len = (float)atoi(arg);
		o->mapQ_coef_fac = o->mapQ_coef_len > 0 ? (int)OPT_LOG(o->mapQ_coef_len) : 0;
		break;
	case 'O': opt_pair(arg, &o->o_del, &o->o_ins); s->o_del = s->o_ins = 1; break;
	case 'E': opt_pair(arg, &o->e_del, &o->e_ins); s->e_del = s->e_ins = 1; break;
	case 'L': opt_pair(arg, &o->pen_clip5, &o->pen_clip3); s->pen_clip5 = s->pen_clip3 = 1; break;
	case 'R': {                                              /* keep the ID: value, as bwa_set_rg does */
		const char *p = strstr(arg, "ID:"), *q;
		if (p == 0) return 1;
		p += 3;
		for (q = p; *q && *q != '\t' && *q != '\n' && !(*q == '\\' && q[1] == 't'); ++q);
		if (q - p >= (long)sizeof(x->rg_id)) return 1;
		memcpy(x->rg_id, p, q - p); x->rg_id[q - p] = 0;
		break;
	}
	case 'I': {                                              /* FLOAT[,FLOAT[,INT[,INT]]]: mean, sd, max, min; FR only (fastmap.c:156-173) */
		char *p;
		PES_T *r = &x->pes[1];
		x->has_pes0 = 1;
		r->failed = 0;
		r->avg = strtod(arg, &p);
		r->std = r->avg * .1;
		if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) r->std = strtod(p + 1, &p);
		r->high = (int)(r->avg + 4. * r->std + .499);
		r->low = (int)(r->avg - 4. * r->std + .499);
		if (r->low < 1) r->low = 1;
		if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) r->high = (int)(strtod(p + 1, &p) + .499);
		if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) r->low = (int)(strtod(p + 1, &p) + .499);
		break;
	}
	default: return 1;
	}
	return 0;
}

/* after the loop: presets (-x), scaling by -A of everything not given explicitly, score matrix */
static int optparse_finish(optparse_t *x)
{
	OPT_T *o = x->opt; const OPT_T *s = &x->set;
	if (o->n_threads < 1) o->n_threads = 1;
	if (x->mode) {
		if (strcmp(x->mode, "intractg") == 0) {
			if (!s->o_del) o->o_del = 16;
			if (!s->o_ins) o->o_ins = 16;
			if (!s->b) o->b = 9;
			if (!s->pen_clip5) o->pen_clip5 = 5;
			if (!s->pen_clip3) o->pen_clip3 = 5;
		} else if (strcmp(x->mode, "pacbio") == 0 || strcmp(x->mode, "pbref") == 0 || strcmp(x->mode, "ont2d") == 0) {
			const int ont = strcmp(x->mode, "ont2d") == 0;
			if (!s->o_del) o->o_del = 1;
			if (!s->e_del) o->e_del = 1;
			if (!s->o_ins) o->o_ins = 1;
			if (!s->e_ins) o->e_ins = 1;
			if (!s->b) o->b = 1;
			if (s->split_factor == 0.f) o->split_factor = 10.f;
			if (!s->min_chain_weight) o->min_chain_weight = ont ? 20 : 40;
			if (!s->min_seed_len) o->min_seed_len = ont ? 14 : 17;
			if (!s->pen_clip5) o->pen_clip5 = 0;
			if (!s->pen_clip3) o->pen_clip3 = 0;
		} else return 1;
	} else if (s->a) {                                       /* the match score was changed: scale what was not set */
		if (!s->b) o->b *= o->a;
		if (!s->T) o->T *= o->a;
		if (!s->o_del) o->o_del *= o->a;
		if (!s->e_del) o->e_del *= o->a;
		if (!s->o_ins) o->o_ins *= o->a;
		if (!s->e_ins) o->e_ins *= o->a;
		if (!s->zdrop) o->zdrop *= o->a;
		if (!s->pen_clip5) o->pen_clip5 *= o->a;
		if (!s->pen_clip3) o->pen_clip3 *= o->a;
		if (!s->pen_unpaired) o->pen_unpaired *= o->a;
	}
	OPT_FILL_SCMAT(o->a, o->b, o->mat);
	return 0;
}

#endif
