/* libbwamem_hip.so -- the symbol the reference's callers bind: mem_process_seqs() with the reference's exact signature
 * (bwamem.h:69, definition bwamem.c:1215), implemented on top of libbwahip.so (include/bwahip.h).
 *
 * A reference build that wants the MI355X path links this library INSTEAD of compiling bwamem.c's definition (see
 * INTEGRATION.md section 1): every call site (the upstream kt_pipeline `process` step, fastmap.c; bwamem-lite's
 * example.c) stays as it is.  The struct types below are layout mirrors of the reference's (checked by
 * tests/test_abi.py against the reference headers), so the caller's mem_opt_t / bwt_t / bntseq_t / bseq1_t /
 * mem_pestat_t objects are passed straight through.
 *
 * Semantics kept from the reference: void return; seqs[i].seq overwritten with 0..4 codes; seqs[i].sam = malloc()ed
 * NUL-terminated SAM text the caller frees; `pes0` NULL => insert-size statistics inferred per batch; errors are fatal
 * (err_fatal style: message on stderr, exit(EXIT_FAILURE), utils.c:90-99).  The index is uploaded to the GPU on the
 * first call and reused while the same (bwt, bns, pac) pointers are passed; BWAHIP_DEVICE picks the HIP device.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/bwamem_hip.h"

static struct { const void *bwt, *bns, *pac; bwahip_ctx *ctx; int atexit_set; } g_cache;

static void compat_release(void)
{
	if (g_cache.ctx) bwahip_destroy(g_cache.ctx);
	g_cache.ctx = 0; g_cache.bwt = g_cache.bns = g_cache.pac = 0;
}

static void compat_fatal(const char *what, int rc)
{
	fprintf(stderr, "[mem_process_seqs] %s failed (bwahip error %d). Abort!\n", what, rc);
	exit(EXIT_FAILURE);
}

/* Read group id (the reference's global bwa_rg_id, bwa.c:44): a caller that honours -R forwards it here. */
static char *g_rg_id = 0;                                  /* own copy: the caller's buffer (bwa_rg_id, bwa.c:44) may be reused */
void bwahip_compat_set_rg_id(const char *id)
{
	char *copy = id && *id ? strdup(id) : 0;
	free(g_rg_id);
	g_rg_id = copy;
	if (g_cache.ctx) bwahip_ctx_set_rg_id(g_cache.ctx, g_rg_id ? g_rg_id : "");
}

/* Drop the cached context (e.g. before unloading the index). */
void bwahip_compat_release(void) { compat_release(); free(g_rg_id); g_rg_id = 0; }

void mem_process_seqs(const bwahip_opt_t *opt, const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac,
                      int64_t n_processed, int n, bwahip_seq_t *seqs, const bwahip_pestat_t *pes0)
{
	int rc;
	if (g_cache.ctx == 0 || g_cache.bwt != (const void*)bwt || g_cache.bns != (const void*)bns || g_cache.pac != (const void*)pac) {
		const char *dev = getenv("BWAHIP_DEVICE");
		compat_release();
		rc = bwahip_init(bwt, bns, pac, dev ? atoi(dev) : 0, &g_cache.ctx);
		if (rc) compat_fatal("bwahip_init", rc);
		g_cache.bwt = bwt; g_cache.bns = bns; g_cache.pac = pac;
		bwahip_ctx_set_rg_id(g_cache.ctx, g_rg_id ? g_rg_id : "");
		if (!g_cache.atexit_set) { atexit(compat_release); g_cache.atexit_set = 1; }
	}
	rc = bwahip_process_seqs(g_cache.ctx, opt, n_processed, n, seqs, pes0);
	if (rc) compat_fatal("bwahip_process_seqs", rc);
}
