// Batch FASTA/FASTQ reader for bwahip_process_seqs: what bseq_read (bwa.c:191) + kseq_read (kseq.h:176) give the
// reference, restated for throughput -- one GPU consumes 10-12 M reads/s, so the reader is a parallel pipeline per input file:
//
//   source thread   raw text in order, in chunks of CHUNK bytes: slices of the mmap()ed file (plain files), the output of zlib
//                   inflate running ahead on this thread (gzip; stdin), or BGZF blocks (bgzip files: self-delimiting members,
//                   found without inflating) whose inflation is part of the parse job, i.e. runs on all workers at once;
//   parse workers   (shared by both files) take chunks in any order: find the first record start in the chunk, parse the
//                   four-line records lying whole inside it -- four memchr + four memcpy each into a Block (text slab + record
//                   table, no malloc per read) -- and leave the bytes before the first start and after the last whole record alone;
//   merger thread   walks the chunks in order: tail of chunk k-1 + head of chunk k must parse exactly into whole four-line
//                   records -- that proves chunk k's speculative first record start was a true one, so by induction the blocks
//                   are what a sequential parse gives.  Whenever that fails (multi-line records, FASTA, CRLF, blank lines, a
//                   truncated last record ...) the file continues from the last verified record boundary through read_one, a
//                   step-by-step restatement of kseq_read, on this thread.  Blocks go to the consumer's queue in file order.
//
// bwahip_fastq_next_batch links records of the block queues into a bseq1_t array (mates interleaved for two files) until the
// batch holds chunk_bases bases and an even number of reads (bwa.c:216); a batch is an owned object holding references on
// its blocks, so several batches can be alive at once (two contexts in flight).  Record syntax as kseq_read: header '>' or '@',
// name up to the first white space, the rest of the line is the comment, sequence over any number of lines up to '+', '>' or
// '@', quality lines until as long as the sequence; "\r\n" line ends; a truncated last record ends the input; trailing
// "/[0-9]" of a name is cut (trim_readno, bwa.c:73).  A read error of the input (corrupt or truncated gzip data) is an error
// of the batch call (BWAHIP_EIO), as err_gzread (utils.c:142) makes it fatal in the reference.
#include "../../include/bwahip.h"
#include <zlib.h>
#include <ctype.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t NONE = 0xffffffffu;
constexpr size_t CHUNK = 4u << 20;                                // raw text per parse job
constexpr size_t BLOCK_RECS = 1 << 16;                            // records per block on the sequential path
struct Rec { uint32_t name, comment, seq, qual; int l_seq; };     // offsets into the block's slab; NONE: no comment / no quality
struct Block { std::vector<char> slab; std::vector<Rec> recs; size_t next = 0; };   // next: first record not yet handed out

// blocks whose batch is gone are reused, so their pages are touched once
struct BlockPool {
	std::mutex mu;
	std::vector<Block*> free_;
	size_t cap = 64;
	Block *get()
	{
		{ std::lock_guard<std::mutex> lk(mu); if (!free_.empty()) { Block *b = free_.back(); free_.pop_back(); b->slab.clear(); b->recs.clear(); b->next = 0; return b; } }
		return new Block();
	}
	void put(Block *b) { std::lock_guard<std::mutex> lk(mu); if (free_.size() < cap) free_.push_back(b); else delete b; }
	~BlockPool() { for (Block *b : free_) delete b; }
};
typedef std::shared_ptr<Block> BlockRef;
static BlockRef new_block(const std::shared_ptr<BlockPool> &pool)
{
	return BlockRef(pool->get(), [pool](Block *x) { pool->put(x); });   // the deleter keeps the pool alive as long as any block is out
}

// One four-line FASTQ record lying whole in [p, e), LF line ends: appended to b, its length returned; 0 = not that layout (multi-line
// record, FASTA, CRLF, a record cut by e, anything else): the caller's business.  Gives the same record as read_one below.
static inline size_t parse_strict(const unsigned char *p, const unsigned char *e, Block &b)
{
	if (p >= e || *p != '@') return 0;
	const unsigned char *n1 = (const unsigned char*)memchr(p, '\n', e - p);
	if (!n1 || n1 + 1 >= e) return 0;
	const unsigned char *n2 = (const unsigned char*)memchr(n1 + 1, '\n', e - (n1 + 1));
	if (!n2 || n2 + 1 >= e || n2[1] != '+') return 0;
	const unsigned char *n3 = (const unsigned char*)memchr(n2 + 1, '\n', e - (n2 + 1));
	if (!n3 || n3 + 1 > e) return 0;
	const unsigned char *n4 = (const unsigned char*)memchr(n3 + 1, '\n', e - (n3 + 1));
	if (!n4) return 0;
	const size_t l_seq = (size_t)(n2 - (n1 + 1));
	const unsigned char *sq = n1 + 1, *ql = n3 + 1;
	if (l_seq == 0 || (size_t)(n4 - ql) != l_seq || *sq == '>' || *sq == '+' || *sq == '@') return 0;
	if (n1[-1] == '\r' || n2[-1] == '\r' || n3[-1] == '\r' || n4[-1] == '\r') return 0;
	const unsigned char *h = p + 1, *sp = h;
	while (sp < n1 && !isspace(*sp)) ++sp;                        // name: up to the first white space (kseq.h:186)
	size_t l_name = (size_t)(sp - h);
	const size_t l_com = sp < n1 ? (size_t)(n1 - (sp + 1)) : 0;   // the rest of the line after that one character (kseq.h:187)
	if (l_name > 2 && h[l_name - 2] == '/' && isdigit(h[l_name - 1])) l_name -= 2;   // trim_readno (bwa.c:73)
	std::vector<char> &t = b.slab;
	const size_t at = t.size();
	if (at + (size_t)(n4 - p) + 8 > 0xfff00000u) return 0;        // offsets are 32 bits
	t.resize(at + l_name + 1 + (l_com ? l_com + 1 : 0) + 2 * (l_seq + 1));
	char *d = t.data() + at;
	Rec r; r.comment = NONE;
	r.name = (uint32_t)at; memcpy(d, h, l_name); d[l_name] = 0; d += l_name + 1;
	if (l_com) { r.comment = (uint32_t)(d - t.data()); memcpy(d, sp + 1, l_com); d[l_com] = 0; d += l_com + 1; }
	r.seq = (uint32_t)(d - t.data()); memcpy(d, sq, l_seq); d[l_seq] = 0; d += l_seq + 1;
	r.qual = (uint32_t)(d - t.data()); memcpy(d, ql, l_seq); d[l_seq] = 0;
	r.l_seq = (int)l_seq;
	b.recs.push_back(r);
	return (size_t)(n4 + 1 - p);
}

// First position >= 0 of [p, p + n) that looks like the start of a four-line record: a line starting with '@' whose next-but-one
// line starts with '+' (a quality line that starts with '@' is followed by a header and a sequence line, never by "x\n+").
// n when there is none.  Speculative: the merger verifies it.  `at_line_start`: p itself is known to start a line.
static size_t resync(const unsigned char *p, size_t n, bool at_line_start)
{
	const unsigned char *e = p + n, *s = p;
	if (!at_line_start) { const unsigned char *nl = (const unsigned char*)memchr(p, '\n', n); if (!nl) return n; s = nl + 1; }
	while (s < e) {
		const unsigned char *n1 = (const unsigned char*)memchr(s, '\n', e - s);
		if (!n1) return n;
		if (*s == '@') {
			const unsigned char *n2 = n1 + 1 < e ? (const unsigned char*)memchr(n1 + 1, '\n', e - (n1 + 1)) : nullptr;
			if (!n2 || n2 + 1 >= e) return n;
			if (n2[1] == '+' && n1[1] != '@' && n1[1] != '+' && n1[1] != '>') return (size_t)(s - p);
		}
		s = n1 + 1;
	}
	return n;
}

// ---- raw text chunks -------------------------------------------------------------------------------------------------
struct Mapping { void *p = nullptr; size_t n = 0; ~Mapping() { if (p && n) munmap(p, n); } };
struct Chunk {
	int64_t id = 0;
	const unsigned char *p = nullptr; size_t n = 0;               // the text (after inflate_members for BGZF chunks)
	std::vector<unsigned char> own;                               // inflated text (gzip / BGZF); empty for mmap slices
	const unsigned char *z = nullptr; size_t zn = 0;             // BGZF: compressed members to inflate (slice of the mapping)
	bool last = false;                                            // end of the input
	// set by the parse job
	BlockRef interior;
	size_t first = 0, tail_beg = 0;
	bool done = false, bad = false;                               // bad: inflate error inside the job
};
typedef std::shared_ptr<Chunk> ChunkRef;

// BGZF members [z, z + zn) -> c.own (every member: 18-byte header with the BC subfield, raw deflate, crc32 + isize)
static bool inflate_members(Chunk &c)
{
	size_t total = 0;
	for (size_t o = 0; o + 18 <= c.zn; ) { const size_t bs = (size_t)(c.z[o + 16] | c.z[o + 17] << 8) + 1; if (o + bs > c.zn || bs < 26) return false; const unsigned char *t = c.z + o + bs - 4; total += (size_t)t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24; o += bs; }
	c.own.resize(total);
	z_stream zs; memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, -15) != Z_OK) return false;
	size_t out = 0; bool ok = true;
	for (size_t o = 0; o + 18 <= c.zn && ok; ) {
		const size_t bs = (size_t)(c.z[o + 16] | c.z[o + 17] << 8) + 1;
		const size_t xlen = (size_t)(c.z[o + 10] | c.z[o + 11] << 8);
		const unsigned char *t = c.z + o + bs - 8;
		const uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
		const size_t isz = (size_t)t[4] | (size_t)t[5] << 8 | (size_t)t[6] << 16 | (size_t)t[7] << 24;
		if (12 + xlen + 8 > bs || out + isz > total) { ok = false; break; }
		zs.next_in = const_cast<unsigned char*>(c.z + o + 12 + xlen); zs.avail_in = (uInt)(bs - 12 - xlen - 8);
		zs.next_out = c.own.data() + out; zs.avail_out = (uInt)isz;
		const int rc = inflate(&zs, Z_FINISH);
		if (rc != Z_STREAM_END || zs.avail_out != 0 || (uint32_t)crc32(0L, c.own.data() + out, (uInt)isz) != crc) { ok = false; break; }
		out += isz;
		inflateReset(&zs);
		o += bs;
	}
	inflateEnd(&zs);
	if (!ok) return false;
	c.p = c.own.data(); c.n = out;
	return true;
}

// ---- the worker pool (shared by the files of one reader) ---------------------------------------------------------------
struct Pool {
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::function<void()>> jobs;
	std::vector<std::thread> th;
	bool stop = false;
	explicit Pool(int n)
	{
		for (int i = 0; i < n; ++i) th.emplace_back([this] {
			for (;;) {
				std::function<void()> f;
				{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !jobs.empty(); }); if (jobs.empty()) return; f = std::move(jobs.front()); jobs.pop_front(); }
				f();
			}
		});
	}
	void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); jobs.push_back(std::move(f)); } cv.notify_one(); }
	~Pool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
};

// ---- one input file --------------------------------------------------------------------------------------------------
struct FileReader {
	// configuration
	std::string path;
	int fd = -1;
	gzFile gz = nullptr;                                          // gzip / stdin
	std::shared_ptr<Mapping> map;                                 // plain or BGZF file
	bool bgzf = false;
	Pool *pool = nullptr;
	std::shared_ptr<BlockPool> blocks;
	size_t max_inflight = 8;
	// source -> merger
	std::mutex mu;
	std::condition_variable cv_chunk, cv_room, cv_out, cv_out_room;
	std::deque<ChunkRef> chunks;                                  // in file order; front = the next one the merger takes
	size_t inflight = 0;
	bool src_done = false, stop = false;
	std::atomic<bool> sequential{false};                         // the rest of the file goes through read_one: parse jobs return at once
	std::atomic<int> error{0};
	// merger -> consumer
	std::deque<BlockRef> out;
	bool out_done = false;
	static constexpr size_t MAX_OUT = 24;
	std::thread th_src, th_merge;

	void fail(const char *what) { if (!error.exchange(1)) fprintf(stderr, "[E::bwahip_fastq] %s: %s\n", path.c_str(), what); }

	// ---------------- source
	bool put_chunk(const ChunkRef &c)                             // false: the reader is being closed
	{
		{
			std::unique_lock<std::mutex> lk(mu);
			cv_room.wait(lk, [&] { return stop || inflight < max_inflight; });
			if (stop) return false;
			++inflight;
			chunks.push_back(c);
		}
		FileReader *self = this;
		pool->push([self, c] { self->parse_job(*c); });
		return true;
	}
	void source()
	{
		int64_t id = 0;
		if (map && !bgzf) {                                        // plain file: slices of the mapping
			const unsigned char *base = (const unsigned char*)map->p;
			const size_t n = map->n;
			for (size_t o = 0; o < n || id == 0; o += CHUNK) {
				ChunkRef c(new Chunk());
				c->id = id++; c->p = base + o; c->n = n - o < CHUNK ? n - o : CHUNK; c->last = o + CHUNK >= n;
				if (!put_chunk(c) || c->last) break;
			}
		} else if (map && bgzf) {                                  // bgzip file: groups of whole members; a foreign member switches to zlib from there
			const unsigned char *z = (const unsigned char*)map->p;
			const size_t n = map->n;
			size_t o = 0;
			bool to_zlib = false;
			while (o < n) {
				size_t e = o, raw = 0;
				while (e + 18 <= n && raw < CHUNK) {
					if (!(z[e] == 0x1f && z[e + 1] == 0x8b && z[e + 2] == 8 && (z[e + 3] & 4) && z[e + 10] == 6 && z[e + 11] == 0 && z[e + 12] == 'B' && z[e + 13] == 'C')) { to_zlib = true; break; }
					const size_t bs = (size_t)(z[e + 16] | z[e + 17] << 8) + 1;
					if (bs < 26 || e + bs > n) { to_zlib = true; break; }
					const unsigned char *t = z + e + bs - 4;
					raw += (size_t)t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24;
					e += bs;
				}
				if (e == o) { if (!to_zlib) fail("truncated BGZF block"); break; }
				ChunkRef c(new Chunk());
				c->id = id++; c->z = z + o; c->zn = e - o;
				o = e;
				if (!put_chunk(c)) return;
				if (to_zlib) break;
			}
			if (to_zlib && o < n) {                                   // not a BGZF member: the rest through zlib's streaming reader
				if (lseek(fd, (off_t)o, SEEK_SET) < 0 || !(gz = gzdopen(dup(fd), "r"))) fail("cannot continue after the BGZF blocks");
				else { gzbuffer(gz, 1 << 20); source_gz(id); return; }
			}
			{ ChunkRef c(new Chunk()); c->id = id++; c->last = true; static const unsigned char nothing = 0; c->p = &nothing; c->n = 0; put_chunk(c); }   // end marker (a group cannot know whether a zlib tail follows it)
		} else source_gz(id);
		{ std::lock_guard<std::mutex> lk(mu); src_done = true; }
		cv_chunk.notify_all();
	}
	void source_gz(int64_t id)                                    // inflate ahead on this thread (plain data passes through gzread unchanged)
	{
		for (;;) {
			ChunkRef c(new Chunk());
			c->own.resize(CHUNK);
			size_t got = 0;
			while (got < CHUNK) {
				const int n = gzread(gz, c->own.data() + got, (unsigned)(CHUNK - got));
				if (n < 0) { int e = 0; const char *m = gzerror(gz, &e); fail(m && *m ? m : "gzread failed"); break; }
				if (n == 0) break;
				got += (size_t)n;
			}
			if (got < CHUNK && !error) {                             // end of the stream: a truncated gzip file shows as Z_BUF_ERROR here only
				int e = 0; const char *m = gzerror(gz, &e);
				if (e != Z_OK && e != Z_STREAM_END) fail(m && *m ? m : "truncated input");
			}
			c->own.resize(got);
			c->id = id++; c->p = c->own.data(); c->n = got; c->last = got < CHUNK || error;
			if (!put_chunk(c) || c->last) break;
		}
		{ std::lock_guard<std::mutex> lk(mu); src_done = true; }
		cv_chunk.notify_all();
	}

	// ---------------- parse job (any worker)
	void parse_job(Chunk &c)
	{
		if (c.z && !inflate_members(c)) { c.bad = true; c.p = (const unsigned char*)""; c.n = 0; }
		if (!sequential && !c.bad && c.n) {
			c.interior = new_block(blocks);
			Block &b = *c.interior;
			b.slab.reserve(c.n + c.n / 16 + 64);
			b.recs.reserve(c.n / 200 + 16);
			const unsigned char *e = c.p + c.n;
			size_t pos = c.first = resync(c.p, c.n, c.id == 0);
			for (size_t len; pos < c.n && (len = parse_strict(c.p + pos, e, b)) != 0; pos += len) {}
			c.tail_beg = pos;
		} else c.first = c.tail_beg = c.n;
		{ std::lock_guard<std::mutex> lk(mu); c.done = true; }
		cv_chunk.notify_all();
	}

	// ---------------- merger
	ChunkRef next_chunk()                                         // the next chunk in file order once its job is done; nullptr at the end / on close
	{
		std::unique_lock<std::mutex> lk(mu);
		cv_chunk.wait(lk, [&] { return stop || (!chunks.empty() && chunks.front()->done) || (chunks.empty() && src_done); });
		if (stop || chunks.empty()) return nullptr;
		ChunkRef c = chunks.front();
		chunks.pop_front();
		--inflight;
		cv_room.notify_all();
		return c;
	}
	bool emit(const BlockRef &b)                                  // false: closing
	{
		if (!b || b->recs.empty()) return true;
		std::unique_lock<std::mutex> lk(mu);
		cv_out_room.wait(lk, [&] { return stop || out.size() < MAX_OUT; });
		if (stop) return false;
		out.push_back(b);
		cv_out.notify_all();
		return true;
	}
	void merge()
	{
		std::vector<unsigned char> carry;                          // unverified bytes that start at a verified record boundary
		bool seq_mode = false;
		ChunkRef c;
		while (!seq_mode && (c = next_chunk())) {
			if (c->bad) { fail("corrupt BGZF block"); break; }
			if (c->first < c->n || c->last) {
				carry.insert(carry.end(), c->p, c->p + c->first);
				bool ok = true;
				if (!carry.empty()) {
					BlockRef b = new_block(blocks);
					size_t pos = 0;
					for (size_t len; pos < carry.size() && (len = parse_strict(carry.data() + pos, carry.data() + carry.size(), *b)) != 0; pos += len) {}
					ok = pos == carry.size();
					if (ok && !emit(b)) return;
				}
				if (!ok) {                                              // continue from the verified boundary through read_one
					sequential = true; seq_mode = true;
					carry.insert(carry.end(), c->p + c->first, c->p + c->n);
					break;
				}
				carry.clear();
				if (!emit(c->interior)) return;
				carry.assign(c->p + c->tail_beg, c->p + c->n);
				if (c->last && !carry.empty()) { sequential = true; seq_mode = true; break; }   // bytes after the last whole four-line record of the file (a truncated or multi-line record, no final newline): read_one's
			} else carry.insert(carry.end(), c->p, c->p + c->n);      // no record start in the whole chunk: part of one long record
			if (c->last) break;
		}
		if (seq_mode) run_sequential(carry, c && c->last);
		{ std::lock_guard<std::mutex> lk(mu); out_done = true; }
		cv_out.notify_all();
	}

	// ---- kseq_read restated step by step over the rest of the input (everything the strict parser does not take)
	struct Stream {
		FileReader *fr; std::vector<unsigned char> buf; size_t beg = 0, end = 0; bool eof = false; ChunkRef cur;
		bool fill()
		{
			if (eof) return false;
			for (;;) {
				cur = fr->next_chunk();
				if (!cur) { eof = true; return false; }
				if (cur->bad) { fr->fail("corrupt BGZF block"); eof = true; return false; }
				if (cur->n) break;
				if (cur->last) { eof = true; return false; }
			}
			buf.assign(cur->p, cur->p + cur->n); beg = 0; end = buf.size();
			if (cur->last) eof_after = true;
			return true;
		}
		bool eof_after = false;
		bool more() { if (beg < end) return true; if (eof_after) { eof = true; return false; } return fill(); }
		int getc() { if (!more()) return -1; return buf[beg++]; }
		// Append bytes up to (not including) the next white space (line == false) or '\n' (line == true) to out; the delimiter is
		// consumed and returned in *dret (0 at the end of the input).  field: where the string being built starts in out -- for lines
		// one trailing '\r' is dropped when that string is longer than one character (kseq.h:140).  false: nothing could be read
		// because the input is exhausted (ks_getuntil2 < 0).
		bool until(bool line, std::vector<char> &o, size_t field, int *dret)
		{
			bool any = false;
			if (dret) *dret = 0;
			for (;;) {
				if (!more()) break;
				size_t i = beg;
				if (line) { const void *p = memchr(buf.data() + beg, '\n', end - beg); i = p ? (size_t)((const unsigned char*)p - buf.data()) : end; }
				else while (i < end && !isspace(buf[i])) ++i;
				any = true;
				o.insert(o.end(), buf.begin() + beg, buf.begin() + i);
				if (i < end) { beg = i + 1; if (dret) *dret = buf[i]; break; }
				beg = end;
			}
			if (!any) return false;
			if (line && o.size() - field > 1 && o.back() == '\r') o.pop_back();
			return true;
		}
	};
	int last_char = 0;
	bool read_one(Stream &s, Block &b)                            // one record appended to b (kseq_read); false at the end of the input (or a truncated record)
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
	void run_sequential(std::vector<unsigned char> &first_bytes, bool first_is_all)
	{
		Stream s; s.fr = this;
		s.buf.swap(first_bytes); s.beg = 0; s.end = s.buf.size(); s.eof_after = first_is_all;
		for (;;) {
			BlockRef b = new_block(blocks);
			bool more = true;
			while (b->recs.size() < BLOCK_RECS && b->slab.size() < (3u << 30) && (more = read_one(s, *b))) {}
			if (!emit(b) || !more) break;
		}
		// drain: nothing is parsed any more, but the source may still be waiting for room
		while (next_chunk()) {}
	}

	// ---------------- consumer
	BlockRef front()                                              // the block holding the next record (nullptr at the end of the input)
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			while (!out.empty() && out.front()->next >= out.front()->recs.size()) { out.pop_front(); cv_out_room.notify_all(); }
			if (!out.empty()) return out.front();
			if (out_done) return nullptr;
			cv_out.wait(lk);
		}
	}
	~FileReader()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv_room.notify_all(); cv_chunk.notify_all(); cv_out.notify_all(); cv_out_room.notify_all();
		if (th_src.joinable()) th_src.join();
		if (th_merge.joinable()) th_merge.join();
		// parse jobs still queued hold their chunks; they finish on the pool (destroyed after the files) and touch only the chunk and mu/cv
	}
	void close_handles() { if (gz) gzclose(gz); gz = nullptr; if (fd >= 0) close(fd); fd = -1; }
};

struct Batch {
	std::vector<bwahip_seq_t> seqs;
	std::vector<BlockRef> held;                                  // the blocks the strings live in
};

} // namespace

struct bwahip_fastq_batch { Batch b; };

struct bwahip_fastq {
	std::unique_ptr<Pool> pool;                                  // declared first: destroyed last (after the files' threads are joined, queued jobs drain here)
	std::shared_ptr<BlockPool> blocks;
	std::unique_ptr<FileReader> fp[2];
	int n_files = 0;
	BlockRef cur[2];                                             // the block records are being taken from, per file (the queue's lock is taken once per block)
	bwahip_fastq_batch *legacy = nullptr;                        // the batch bwahip_fastq_next hands out (released by the next call)
	~bwahip_fastq()
	{
		cur[0].reset(); cur[1].reset();
		delete legacy;
		for (auto &f : fp) if (f) { std::lock_guard<std::mutex> lk(f->mu); f->stop = true; }
		for (auto &f : fp) if (f) { f->cv_room.notify_all(); f->cv_chunk.notify_all(); f->cv_out.notify_all(); f->cv_out_room.notify_all(); }
		for (auto &f : fp) if (f) { if (f->th_src.joinable()) f->th_src.join(); if (f->th_merge.joinable()) f->th_merge.join(); }
		pool.reset();                                             // joins the workers: no job refers to a FileReader after this
		for (auto &f : fp) if (f) { f->out.clear(); f->chunks.clear(); f->close_handles(); }
	}
};

extern "C" {

int bwahip_fastq_open_mt(const char *path1, const char *path2, int n_threads, bwahip_fastq **out)
{
	if (!path1 || !out) return BWAHIP_EINVAL;
	if (n_threads <= 0) {
		if (const char *e = getenv("BWAHIP_READER_THREADS")) n_threads = atoi(e);
		if (n_threads <= 0) { const unsigned hw = std::thread::hardware_concurrency(); n_threads = hw >= 16 ? 8 : hw >= 4 ? (int)hw / 2 : 1; }
	}
	if (n_threads > 64) n_threads = 64;
	std::unique_ptr<bwahip_fastq> r(new bwahip_fastq());
	r->pool.reset(new Pool(n_threads));
	r->blocks.reset(new BlockPool());
	r->blocks->cap = 256;
	const char *paths[2] = { path1, path2 };
	r->n_files = path2 ? 2 : 1;
	for (int k = 0; k < r->n_files; ++k) {
		r->fp[k].reset(new FileReader());
		FileReader &p = *r->fp[k];
		p.path = paths[k]; p.pool = r->pool.get(); p.blocks = r->blocks;
		// how far the parser may run ahead of the consumer, per file: BWAHIP_READER_AHEAD_MB of raw text (default 72 MB with 8 threads: 2n + 2
		// jobs of 4 MB; a driver with several contexts asks for batches in bursts and can set a whole batch's worth)
		p.max_inflight = (size_t)(2 * n_threads + 2);
		if (const char *e = getenv("BWAHIP_READER_AHEAD_MB")) { const long mb = atol(e); if (mb > 0) p.max_inflight = std::max<size_t>((size_t)mb * (1u << 20) / CHUNK, 2); }
		if (strcmp(paths[k], "-") == 0) { p.gz = gzdopen(dup(0), "r"); if (!p.gz) return BWAHIP_EIO; gzbuffer(p.gz, 1 << 20); continue; }
		p.fd = open(paths[k], O_RDONLY);
		if (p.fd < 0) { fprintf(stderr, "[bwahip] cannot open %s\n", paths[k]); return BWAHIP_EIO; }
		struct stat st;
		unsigned char magic[18]; ssize_t got = 0;
		const bool regular = fstat(p.fd, &st) == 0 && S_ISREG(st.st_mode);
		if (regular) got = pread(p.fd, magic, sizeof magic, 0);
		const bool is_gz = got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b;
		const bool is_bgzf = is_gz && got >= 18 && magic[2] == 8 && (magic[3] & 4) && magic[10] == 6 && magic[11] == 0 && magic[12] == 'B' && magic[13] == 'C';
		if (regular && st.st_size > 0 && (!is_gz || is_bgzf) && !getenv("BWAHIP_READER_NO_MMAP")) {
			void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, p.fd, 0);
			if (m != MAP_FAILED) {
				(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
				p.map.reset(new Mapping()); p.map->p = m; p.map->n = (size_t)st.st_size; p.bgzf = is_bgzf;
				continue;
			}
		}
		p.gz = gzdopen(dup(p.fd), "r");                            // gzip, an empty file, a pipe: zlib's reader (plain data passes through)
		if (!p.gz) { fprintf(stderr, "[bwahip] cannot open %s\n", paths[k]); return BWAHIP_EIO; }
		gzbuffer(p.gz, 1 << 20);
	}
	for (int k = 0; k < r->n_files; ++k) {
		FileReader *p = r->fp[k].get();
		p->th_src = std::thread([p] { p->source(); });
		p->th_merge = std::thread([p] { p->merge(); });
	}
	*out = r.release();
	return 0;
}

int bwahip_fastq_open(const char *path1, const char *path2, bwahip_fastq **out) { return bwahip_fastq_open_mt(path1, path2, 0, out); }

// bseq_read (bwa.c:191): the next batch as an owned object.  *n = 0 (and *batch = NULL) at the end of the input.
int bwahip_fastq_next_batch(bwahip_fastq *r, int64_t chunk_bases, int keep_comments, bwahip_fastq_batch **batch, bwahip_seq_t **seqs, int *n)
{
	if (!r || !batch || !n) return BWAHIP_EINVAL;
	*batch = nullptr; *n = 0; if (seqs) *seqs = nullptr;
	std::unique_ptr<bwahip_fastq_batch> bt(new bwahip_fastq_batch());
	Batch &b = bt->b;
	for (int k = 0; k < r->n_files; ++k) if (r->cur[k] && r->cur[k]->next < r->cur[k]->recs.size()) b.held.push_back(r->cur[k]);   // a block carried over from the last batch
	int64_t size = 0;
	auto take = [&](int k) -> bool {
		if (!r->cur[k] || r->cur[k]->next >= r->cur[k]->recs.size()) {
			r->cur[k] = r->fp[k]->front();
			if (!r->cur[k]) return false;
			b.held.push_back(r->cur[k]);
		}
		Block *bl = r->cur[k].get();
		const Rec &rec = bl->recs[bl->next++];
		bwahip_seq_t s;
		memset(&s, 0, sizeof s);
		char *base = bl->slab.data();
		s.name = base + rec.name;
		s.comment = keep_comments && rec.comment != NONE ? base + rec.comment : nullptr;
		s.seq = base + rec.seq;
		s.qual = rec.qual != NONE ? base + rec.qual : nullptr;
		s.l_seq = rec.l_seq;
		s.id = (int)b.seqs.size();
		b.seqs.push_back(s);
		size += rec.l_seq;
		return true;
	};
	if (b.seqs.capacity() == 0) b.seqs.reserve(chunk_bases > 0 && chunk_bases < (1ll << 32) ? (size_t)(chunk_bases / 100 + 16) : 1024);
	for (;;) {
		if (!take(0)) break;
		if (r->n_files == 2 && !take(1)) {
			fprintf(stderr, "[W::bwahip_fastq_next] the 2nd file has fewer sequences.\n");
			b.seqs.pop_back();                                    // bseq_read breaks before storing the unpaired first mate
			break;
		}
		if (size >= chunk_bases && (b.seqs.size() & 1) == 0) break;
	}
	for (int k = 0; k < r->n_files; ++k) if (r->fp[k]->error) return BWAHIP_EIO;   // err_gzread (utils.c:142): a damaged input is fatal, never a short batch
	if (b.seqs.empty()) return 0;
	*n = (int)b.seqs.size();
	if (seqs) *seqs = b.seqs.data();
	*batch = bt.release();
	return 0;
}

bwahip_seq_t *bwahip_fastq_batch_seqs(bwahip_fastq_batch *b, int *n)
{
	if (!b) { if (n) *n = 0; return nullptr; }
	if (n) *n = (int)b->b.seqs.size();
	return b->b.seqs.data();
}

void bwahip_fastq_batch_release(bwahip_fastq_batch *b) { delete b; }

// The same with the batch kept by the reader: its strings stay valid until the next call or bwahip_fastq_close.
int bwahip_fastq_next(bwahip_fastq *r, int64_t chunk_bases, int keep_comments, bwahip_seq_t **seqs, int *n)
{
	if (!r || !seqs || !n) return BWAHIP_EINVAL;
	delete r->legacy; r->legacy = nullptr;
	static bwahip_seq_t none;
	const int rc = bwahip_fastq_next_batch(r, chunk_bases, keep_comments, &r->legacy, seqs, n);
	if (!rc && *n == 0) *seqs = &none;
	return rc;
}

void bwahip_fastq_close(bwahip_fastq *r) { delete r; }

} // extern "C"
