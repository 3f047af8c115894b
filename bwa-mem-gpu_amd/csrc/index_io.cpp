// Host-side reader for stock `bwa index` file sets (product code; the boundary's data format).
// Layouts: .bwt primary,L2[1..4],words (bwt.c:385-393, 443-462); .sa primary,4 skipped,sa_intv,seq_len,
// n_sa-1 values (bwt.c:396-441); .pac (bntseq.c:314-326, read as l_pac/4+1 bytes like bwa.c:421);
// .ann/.amb text (bntseq.c:65-94, 97-166); optional .alt list of ALT contig names (bntseq.c:178-209).
#include "bwahip_internal.h"
#include <errno.h>

namespace {

struct File {
	FILE *fp = nullptr;
	File(const std::string &fn, const char *mode) { fp = fopen(fn.c_str(), mode); }
	~File() { if (fp) fclose(fp); }
	bool read(void *p, size_t sz, size_t n) { return fread(p, sz, n, fp) == n; }
};

bool fail(const char *what, const std::string &fn)
{
	fprintf(stderr, "[bwahip] index: %s: %s\n", what, fn.c_str());
	return false;
}

bool load_bwt(const std::string &prefix, bwahip_bwt_t *b)
{
	memset(b, 0, sizeof(*b));
	{
		File f(prefix + ".bwt", "rb");
		if (!f.fp) return fail("cannot open", prefix + ".bwt");
		fseek(f.fp, 0, SEEK_END);
		long sz = ftell(f.fp);
		fseek(f.fp, 0, SEEK_SET);
		if (sz < 40 || ((sz - 40) & 3)) return fail("bad size", prefix + ".bwt");
		b->bwt_size = (uint64_t)(sz - 40) >> 2;
		if (!f.read(&b->primary, 8, 1) || !f.read(b->L2 + 1, 8, 4)) return fail("short read", prefix + ".bwt");
		b->bwt = (uint32_t*)malloc(b->bwt_size * 4);
		if (!b->bwt || !f.read(b->bwt, 4, b->bwt_size)) return fail("short read", prefix + ".bwt");
		b->seq_len = b->L2[4];
	}
	{
		File f(prefix + ".sa", "rb");
		uint64_t hdr[5], v[2];
		if (!f.fp) return fail("cannot open", prefix + ".sa");
		if (!f.read(hdr, 8, 5) || !f.read(v, 8, 2)) return fail("short read", prefix + ".sa");
		if (hdr[0] != b->primary || v[1] != b->seq_len) return fail("SA-BWT inconsistency", prefix + ".sa");
		b->sa_intv = (int)v[0];
		if (b->sa_intv <= 0 || (b->sa_intv & (b->sa_intv - 1))) return fail("SA interval is not a power of 2", prefix + ".sa");
		b->n_sa = (b->seq_len + b->sa_intv) / b->sa_intv;
		b->sa = (uint64_t*)malloc(b->n_sa * 8);
		if (!b->sa) return fail("out of memory", prefix + ".sa");
		b->sa[0] = (uint64_t)-1;
		if (!f.read(b->sa + 1, 8, b->n_sa - 1)) return fail("short read", prefix + ".sa");
	}
	return true;
}

bool load_bns(const std::string &prefix, bwahip_bns_t *bns, uint8_t **pac)
{
	memset(bns, 0, sizeof(*bns));
	char buf[8192];
	long long xx;
	{
		File f(prefix + ".ann", "r");
		if (!f.fp) return fail("cannot open", prefix + ".ann");
		if (fscanf(f.fp, "%lld%d%u", &xx, &bns->n_seqs, &bns->seed) != 3) return fail("parse error", prefix + ".ann");
		bns->l_pac = xx;
		bns->anns = (bwahip_ann_t*)calloc(bns->n_seqs > 0 ? bns->n_seqs : 1, sizeof(bwahip_ann_t));
		for (int i = 0; i < bns->n_seqs; ++i) {
			bwahip_ann_t *p = &bns->anns[i];
			char *q = buf;
			int c = 0;
			if (fscanf(f.fp, "%u%8191s", &p->gi, buf) != 2) return fail("parse error", prefix + ".ann");
			p->name = strdup(buf);
			while (q - buf < (long)sizeof(buf) - 1 && (c = fgetc(f.fp)) != '\n' && c != EOF) *q++ = (char)c;
			while (c != '\n' && c != EOF) c = fgetc(f.fp);
			*q = 0;
			p->anno = (q - buf > 1 && strcmp(buf, " (null)") != 0) ? strdup(buf + 1) : strdup("");
			if (fscanf(f.fp, "%lld%d%d", &xx, &p->len, &p->n_ambs) != 3) return fail("parse error", prefix + ".ann");
			p->offset = xx;
		}
	}
	{
		File f(prefix + ".amb", "r");
		int n_seqs;
		if (!f.fp) return fail("cannot open", prefix + ".amb");
		if (fscanf(f.fp, "%lld%d%d", &xx, &n_seqs, &bns->n_holes) != 3) return fail("parse error", prefix + ".amb");
		if (xx != bns->l_pac || n_seqs != bns->n_seqs) return fail("inconsistent with .ann", prefix + ".amb");
		bns->ambs = bns->n_holes ? (bwahip_amb_t*)calloc(bns->n_holes, sizeof(bwahip_amb_t)) : nullptr;
		for (int i = 0; i < bns->n_holes; ++i) {
			if (fscanf(f.fp, "%lld%d%8191s", &xx, &bns->ambs[i].len, buf) != 3) return fail("parse error", prefix + ".amb");
			bns->ambs[i].offset = xx; bns->ambs[i].amb = buf[0];
		}
	}
	{
		File f(prefix + ".pac", "rb");
		if (!f.fp) return fail("cannot open", prefix + ".pac");
		*pac = (uint8_t*)calloc(bns->l_pac / 4 + 1, 1);
		if (!*pac || !f.read(*pac, 1, bns->l_pac / 4 + 1)) return fail("short read", prefix + ".pac");
	}
	{
		File f(prefix + ".alt", "r");
		if (f.fp) {
			while (fgets(buf, sizeof buf, f.fp)) {
				if (buf[0] == '@') continue;
				char *e = buf;
				while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e;
				*e = 0;
				for (int i = 0; i < bns->n_seqs; ++i)
					if (strcmp(bns->anns[i].name, buf) == 0) { bns->anns[i].is_alt = 1; break; }
			}
		}
	}
	return true;
}

} // namespace

int bwahip_load_index_files(const char *prefix, HostIndex *out)
{
	memset(out, 0, sizeof(*out));
	out->owned = true;
	if (!load_bwt(prefix, &out->bwt) || !load_bns(prefix, &out->bns, &out->pac)) {
		bwahip_free_host_index(out);
		return BWAHIP_EIO;
	}
	return 0;
}

// Deep copy of what finalisation needs on the host (contig table with names, holes, packed reference); the big FM-index
// arrays are not copied (bwt.bwt / bwt.sa stay NULL in the copy).  pac == NULL: left to the caller to fill h->pac.
int bwahip_copy_host_index(const bwahip_bwt_t *bwt, const bwahip_bns_t *bns, const uint8_t *pac, HostIndex *out)
{
	memset(out, 0, sizeof(*out));
	out->owned = true;
	out->bwt = *bwt; out->bwt.bwt = nullptr; out->bwt.sa = nullptr;
	out->bns = *bns; out->bns.anns = nullptr; out->bns.ambs = nullptr; out->bns.fp_pac = nullptr;
	out->bns.anns = (bwahip_ann_t*)calloc(bns->n_seqs > 0 ? bns->n_seqs : 1, sizeof(bwahip_ann_t));
	if (!out->bns.anns) return BWAHIP_ENOMEM;
	for (int i = 0; i < bns->n_seqs; ++i) {
		out->bns.anns[i] = bns->anns[i];
		out->bns.anns[i].name = strdup(bns->anns[i].name ? bns->anns[i].name : "");
		out->bns.anns[i].anno = strdup(bns->anns[i].anno ? bns->anns[i].anno : "");
	}
	if (bns->n_holes > 0 && bns->ambs) {
		out->bns.ambs = (bwahip_amb_t*)malloc((size_t)bns->n_holes * sizeof(bwahip_amb_t));
		if (!out->bns.ambs) return BWAHIP_ENOMEM;
		memcpy(out->bns.ambs, bns->ambs, (size_t)bns->n_holes * sizeof(bwahip_amb_t));
	} else out->bns.n_holes = 0;
	out->pac = (uint8_t*)malloc((size_t)bns->l_pac / 4 + 1);
	if (!out->pac) return BWAHIP_ENOMEM;
	if (pac) memcpy(out->pac, pac, (size_t)bns->l_pac / 4 + 1);
	return 0;
}

void bwahip_free_host_index(HostIndex *h)
{
	if (!h || !h->owned) return;
	free(h->bwt.bwt); free(h->bwt.sa);
	if (h->bns.anns) for (int i = 0; i < h->bns.n_seqs; ++i) { free(h->bns.anns[i].name); free(h->bns.anns[i].anno); }
	free(h->bns.anns); free(h->bns.ambs); free(h->pac);
	memset(h, 0, sizeof(*h));
}

// Concatenate the SAM text of seqs[0..n) into one malloc()ed buffer (read order) and free the per-read strings: what a
// caller that writes the batch's SAM with one fwrite wants (the reference's output step, fastmap.c, fputs per read).
extern "C" int bwahip_seqs_take_sam(bwahip_seq_t *seqs, int n, char **out, int64_t *out_len)
{
	if (n < 0 || (n && !seqs) || !out || !out_len) return BWAHIP_EINVAL;
	std::vector<size_t> off((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (seqs[i].sam ? strlen(seqs[i].sam) : 0);
	char *buf = (char*)malloc(off[n] + 1);
	if (!buf) return BWAHIP_ENOMEM;
	{
		const int nt = n >= 4096 ? 16 : 1;
		std::vector<std::thread> th;
		auto work = [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) if (seqs[i].sam) { memcpy(buf + off[i], seqs[i].sam, off[i + 1] - off[i]); free(seqs[i].sam); seqs[i].sam = nullptr; } };
		for (int k = 1; k < nt; ++k) th.emplace_back(work, (int64_t)n * k / nt, (int64_t)n * (k + 1) / nt);
		work(0, (int64_t)n / nt);
		for (auto &x : th) x.join();
	}
	buf[off[n]] = 0;
	*out = buf; *out_len = (int64_t)off[n];
	return 0;
}
