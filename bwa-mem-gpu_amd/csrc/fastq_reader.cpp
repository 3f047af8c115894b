// Batch FASTA/FASTQ reader for bwahip_process_seqs: what bseq_read (bwa.c:191) + kseq_read (kseq.h:176) give the
// reference, restated for throughput.  One reader thread per input file inflates (zlib: plain and gzip files alike) and parses
// ahead of the consumer into blocks -- a text slab plus a record table, no per-read malloc -- so that reading the next batch runs
// under the GPU's work on the current one; bwahip_fastq_next only links records of the block queues into a bseq1_t array
// (mates interleaved for two files) until the batch holds chunk_bases bases and an even number of reads (bwa.c:216).
// Record syntax as kseq_read: header '>' or '@', name up to the first white space, the rest of the line is the comment,
// sequence over any number of lines up to '+', '>' or '@', quality lines until as long as the sequence; "\r\n" line ends;
// a truncated last record ends the input; trailing "/[0-9]" of a name is cut (trim_readno, bwa.c:73).
#include "../../include/bwahip.h"
#include <zlib.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t NONE = 0xffffffffu;
struct Rec { uint32_t name, comment, seq, qual; int l_seq; };     // offsets into the block's slab; NONE: no comment / no quality
struct Block { std::vector<char> slab; std::vector<Rec> recs; size_t next = 0; };   // next: first record not yet handed out

// buffered byte stream over gzread (reads plain files transparently)
struct Stream {
	gzFile f = nullptr;
	std::vector<unsigned char> buf;
	size_t beg = 0, end = 0;
	bool eof = false;
	bool fill()
	{
		if (eof) return false;
		const int n = gzread(f, buf.data(), (unsigned)buf.size());
		beg = 0; end = n > 0 ? (size_t)n : 0;
		if (n <= 0) { eof = true; return false; }
		return true;
	}
	int getc() { if (beg >= end && !fill()) return -1; return buf[beg++]; }
	// Append bytes up to (not including) the next white space (line == false) or '\n' (line == true) to out; the delimiter is
	// consumed and returned in *dret (0 at the end of the input).  field: where the string being built starts in out -- for lines
	// one trailing '\r' is dropped when that string is longer than one character (kseq.h:140).  false: nothing could be read
	// because the input is exhausted (ks_getuntil2 < 0).
	bool until(bool line, std::vector<char> &out, size_t field, int *dret)
	{
		bool any = false;
		if (dret) *dret = 0;
		for (;;) {
			if (beg >= end && !fill()) break;
			size_t i = beg;
			if (line) { const void *p = memchr(buf.data() + beg, '\n', end - beg); i = p ? (size_t)((const unsigned char*)p - buf.data()) : end; }
			else while (i < end && !isspace(buf[i])) ++i;
			any = true;
			out.insert(out.end(), buf.begin() + beg, buf.begin() + i);
			if (i < end) { beg = i + 1; if (dret) *dret = buf[i]; break; }
			beg = end;
		}
		if (!any) return false;
		if (line && out.size() - field > 1 && out.back() == '\r') out.pop_back();
		return true;
	}
};

struct FileParser {
	Stream s;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv_put, cv_get;
	std::deque<std::shared_ptr<Block>> q;
	bool done = false, stop = false;
	int last_char = 0;
	std::vector<Block*> pool;                                   // blocks whose batch is gone: reused, so their pages are touched once
	Block *fresh()
	{
		{ std::lock_guard<std::mutex> lk(mu); if (!pool.empty()) { Block *b = pool.back(); pool.pop_back(); b->slab.clear(); b->recs.clear(); b->next = 0; return b; } }
		Block *b = new Block();
		b->slab.reserve(BLOCK_RECS * 400);
		b->recs.reserve(BLOCK_RECS);
		return b;
	}
	void recycle(Block *b) { std::lock_guard<std::mutex> lk(mu); if (pool.size() < 2 * MAX_Q + 4) pool.push_back(b); else delete b; }
	static constexpr size_t MAX_Q = 6, BLOCK_RECS = 1 << 16;

	// The common layout -- a four-line FASTQ record lying whole in the buffer, LF line ends -- taken with four memchr and four memcpy.
	// Anything else (multi-line records, FASTA, CRLF, a record across the buffer end, a header character already consumed) returns
	// false with nothing consumed and goes through read_one, which restates kseq_read step by step; both give the same records.
	bool read_fast(Block &b)
	{
		if (last_char != 0 || s.beg >= s.end) return false;
		const unsigned char *p = s.buf.data() + s.beg, *e = s.buf.data() + s.end;
		if (*p != '@') return false;
		const unsigned char *n1 = (const unsigned char*)memchr(p, '\n', e - p);
		if (!n1 || n1 + 1 >= e) return false;
		const unsigned char *n2 = (const unsigned char*)memchr(n1 + 1, '\n', e - (n1 + 1));
		if (!n2 || n2 + 1 >= e || n2[1] != '+') return false;
		const unsigned char *n3 = (const unsigned char*)memchr(n2 + 1, '\n', e - (n2 + 1));
		if (!n3 || n3 + 1 >= e) return false;
		const unsigned char *n4 = (const unsigned char*)memchr(n3 + 1, '\n', e - (n3 + 1));
		if (!n4) return false;
		const size_t l_seq = (size_t)(n2 - (n1 + 1));
		const unsigned char *sq = n1 + 1, *ql = n3 + 1;
		if (l_seq == 0 || (size_t)(n4 - ql) != l_seq || *sq == '>' || *sq == '+' || *sq == '@') return false;
		if (n1[-1] == '\r' || n2[-1] == '\r' || n3[-1] == '\r' || n4[-1] == '\r') return false;
		const unsigned char *h = p + 1, *sp = h;
		while (sp < n1 && !isspace(*sp)) ++sp;                    // name: up to the first white space (kseq.h:186)
		size_t l_name = (size_t)(sp - h);
		const size_t l_com = sp < n1 ? (size_t)(n1 - (sp + 1)) : 0;   // the rest of the line after that one character (kseq.h:187)
		if (l_name > 2 && h[l_name - 2] == '/' && isdigit(h[l_name - 1])) l_name -= 2;   // trim_readno (bwa.c:73)
		std::vector<char> &t = b.slab;
		const size_t at = t.size();
		t.resize(at + l_name + 1 + (l_com ? l_com + 1 : 0) + 2 * (l_seq + 1));
		char *d = t.data() + at;
		Rec r; r.comment = NONE;
		r.name = (uint32_t)at; memcpy(d, h, l_name); d[l_name] = 0; d += l_name + 1;
		if (l_com) { r.comment = (uint32_t)(d - t.data()); memcpy(d, sp + 1, l_com); d[l_com] = 0; d += l_com + 1; }
		r.seq = (uint32_t)(d - t.data()); memcpy(d, sq, l_seq); d[l_seq] = 0; d += l_seq + 1;
		r.qual = (uint32_t)(d - t.data()); memcpy(d, ql, l_seq); d[l_seq] = 0;
		r.l_seq = (int)l_seq;
		b.recs.push_back(r);
		s.beg = (size_t)(n4 + 1 - s.buf.data());
		return true;
	}

	// one record appended to b (kseq_read); false at the end of the input (or a truncated record)
	bool read_one(Block &b)
	{
		int c;
		if (last_char == 0) {
			while ((c = s.getc()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return false;
			last_char = c;
		}
		std::vector<char> &t = b.slab;
		const size_t mark = t.size();
		Rec r; r.comment = NONE; r.qual = NONE;
		r.name = (uint32_t)t.size();
		if (!s.until(false, t, r.name, &c)) { t.resize(mark); return false; }
		if (t.size() - r.name > 2 && t[t.size() - 2] == '/' && isdigit((unsigned char)t.back())) t.resize(t.size() - 2);   // trim_readno (bwa.c:73)
		t.push_back(0);
		if (c != '\n') {                                         // the rest of the header line is the comment (kseq.h:187); empty = none (bwa.c:185)
			const size_t cs = t.size();
			s.until(true, t, cs, nullptr);
			if (t.size() > cs) { r.comment = (uint32_t)cs; t.push_back(0); }
		}
		r.seq = (uint32_t)t.size();
		while ((c = s.getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;                              // empty line
			t.push_back((char)c);
			s.until(true, t, r.seq, nullptr);
		}
		last_char = (c == '>' || c == '@') ? c : 0;
		r.l_seq = (int)(t.size() - r.seq);
		t.push_back(0);
		if (c == '+') {
			while ((c = s.getc()) != -1 && c != '\n') {}          // rest of the '+' line
			if (c == -1) { t.resize(mark); return false; }        // kseq_read returns -2: bseq_read stops here
			const size_t qs = t.size();
			while (s.until(true, t, qs, nullptr) && (int)(t.size() - qs) < r.l_seq) {}
			last_char = 0;
			if ((int)(t.size() - qs) != r.l_seq) { t.resize(mark); return false; }
			t.push_back(0);
			if (r.l_seq > 0) r.qual = (uint32_t)qs;               // dupkstring(&ks->qual, 0): no string for an empty quality
		}
		b.recs.push_back(r);
		return true;
	}

	void run()
	{
		for (;;) {
			std::shared_ptr<Block> b(fresh(), [this](Block *x) { recycle(x); });
			bool more = true;
			while (b->recs.size() < BLOCK_RECS && b->slab.size() < (3u << 30) && (read_fast(*b) || (more = read_one(*b)))) {}
			std::unique_lock<std::mutex> lk(mu);
			cv_put.wait(lk, [&] { return q.size() < MAX_Q || stop; });
			if (stop) return;
			if (!b->recs.empty()) q.push_back(b);
			if (!more) done = true;
			cv_get.notify_all();
			if (!more) return;
		}
	}
	// the block holding the next record (nullptr at the end of the input); blocks the caller until the parser has one
	std::shared_ptr<Block> front()
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			while (!q.empty() && q.front()->next >= q.front()->recs.size()) { q.pop_front(); cv_put.notify_all(); }
			if (!q.empty()) return q.front();
			if (done) return nullptr;
			cv_get.wait(lk);
		}
	}
	~FileParser()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv_put.notify_all();
		if (th.joinable()) th.join();
		q.clear();                                               // their deleters put them into the pool
		for (Block *b : pool) delete b;
		pool.clear();
		if (s.f) gzclose(s.f);
	}
};

} // namespace

struct bwahip_fastq {
	std::unique_ptr<FileParser> fp[2];
	int n_files = 0;
	std::vector<bwahip_seq_t> seqs;
	std::vector<std::shared_ptr<Block>> held;                   // blocks the current batch points into
	std::shared_ptr<Block> cur[2];                              // the block records are being taken from, per file (the queue's lock is taken once per block)
};

extern "C" {

int bwahip_fastq_open(const char *path1, const char *path2, bwahip_fastq **out)
{
	if (!path1 || !out) return BWAHIP_EINVAL;
	std::unique_ptr<bwahip_fastq> r(new bwahip_fastq());
	const char *paths[2] = { path1, path2 };
	r->n_files = path2 ? 2 : 1;
	for (int k = 0; k < r->n_files; ++k) {
		r->fp[k].reset(new FileParser());
		FileParser &p = *r->fp[k];
		p.s.f = strcmp(paths[k], "-") == 0 ? gzdopen(0, "r") : gzopen(paths[k], "r");
		if (!p.s.f) { fprintf(stderr, "[bwahip] cannot open %s\n", paths[k]); return BWAHIP_EIO; }
		gzbuffer(p.s.f, 1 << 20);
		p.s.buf.resize(4 << 20);
	}
	for (int k = 0; k < r->n_files; ++k) { FileParser *p = r->fp[k].get(); p->th = std::thread([p] { p->run(); }); }
	*out = r.release();
	return 0;
}

// bseq_read (bwa.c:191): the next batch.  *n = 0 at the end of the input.  The strings of seqs[] live in the reader and stay valid
// until the next call or bwahip_fastq_close; only seqs[i].sam (set by bwahip_process_seqs) is the caller's to free.
int bwahip_fastq_next(bwahip_fastq *r, int64_t chunk_bases, int keep_comments, bwahip_seq_t **seqs, int *n)
{
	if (!r || !seqs || !n) return BWAHIP_EINVAL;
	r->seqs.clear();
	r->held.clear();
	for (int k = 0; k < r->n_files; ++k) if (r->cur[k] && r->cur[k]->next < r->cur[k]->recs.size()) r->held.push_back(r->cur[k]);   // a block carried over from the last batch
	int64_t size = 0;
	auto take = [&](int k) -> bool {
		if (!r->cur[k] || r->cur[k]->next >= r->cur[k]->recs.size()) {
			r->cur[k] = r->fp[k]->front();
			if (!r->cur[k]) return false;
			r->held.push_back(r->cur[k]);
		}
		Block *b = r->cur[k].get();
		const Rec &rec = b->recs[b->next++];
		bwahip_seq_t s;
		memset(&s, 0, sizeof s);
		char *base = b->slab.data();
		s.name = base + rec.name;
		s.comment = keep_comments && rec.comment != NONE ? base + rec.comment : nullptr;
		s.seq = base + rec.seq;
		s.qual = rec.qual != NONE ? base + rec.qual : nullptr;
		s.l_seq = rec.l_seq;
		s.id = (int)r->seqs.size();
		r->seqs.push_back(s);
		size += rec.l_seq;
		return true;
	};
	for (;;) {
		if (!take(0)) break;
		if (r->n_files == 2 && !take(1)) {
			fprintf(stderr, "[W::bwahip_fastq_next] the 2nd file has fewer sequences.\n");
			r->seqs.pop_back();                                   // bseq_read breaks before storing the unpaired first mate
			break;
		}
		if (size >= chunk_bases && (r->seqs.size() & 1) == 0) break;
	}
	*seqs = r->seqs.data();
	*n = (int)r->seqs.size();
	return 0;
}

void bwahip_fastq_close(bwahip_fastq *r)
{
	if (!r) return;
	r->held.clear(); r->cur[0].reset(); r->cur[1].reset();     // the batch's blocks go back before their parsers do
	delete r;
}

} // extern "C"
