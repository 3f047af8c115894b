// The batch driver behind the C ABI: FASTQ files in -> SAM text out, over any number of contexts.  What superBatchMain
// (cuda/superbatch_process.cpp:133: read || process, double buffered, one GPU) and process() / kt_pipeline of fastmap.c
// (read -> mem_process_seqs -> fputs, fastmap.c:46,307) do in the reference, for N GPUs and several batches in flight per GPU:
//
//   one reader      bwahip_fastq_* (its own parse / inflate threads) cuts batches exactly as bseq_read does (-K bases);
//   N workers       one host thread per context (contexts on N devices, or clones sharing one device's index).  A worker takes the
//                   next batch under the reader's lock -- which also fixes the batch's number and its true n_processed (the global
//                   index of its first read: hash_64 tie-breaks, bwamem.c:534/1204, and the per-batch mem_pestat then come out as in
//                   a serial run) -- and runs bwahip_process_seqs_text on its context;
//   one writer      writes the batches' SAM in batch order to the caller's file descriptor while the workers go on (the text of a
//                   context stays valid until its next-but-one call: bwahip_process_seqs_text alternates between two pinned buffers).
//
// Whole batches are dealt to whichever context is free (on equal devices that is round-robin); results do not depend on which
// context took a batch (tests/test_gpu_multi.py).  No data-path collective: SURVEY.md 8(e).
#include "../../include/bwahip.h"
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Driver {
	// reader side
	std::mutex mu_read;
	bwahip_fastq *rd = nullptr;
	int64_t n_processed = 0, next_seq = 0, max_reads = 0;
	bool eof = false;
	// writer side
	std::mutex mu;
	std::condition_variable cv_item, cv_done;
	struct Item { const char *p; int64_t len; };
	std::map<int64_t, Item> ready;                               // finished batches waiting for their turn
	int64_t written = 0;                                         // batches [0, written) are on the descriptor
	int workers_left = 0;
	int rc = 0;                                                  // first error (workers and the writer stop on it)
	int fd = -1;
	int64_t sam_bytes = 0;
	double t_last_write = 0, write_s = 0;

	void fail(int code) { { std::lock_guard<std::mutex> lk(mu); if (!rc) rc = code; } cv_item.notify_all(); cv_done.notify_all(); }
	bool failed() { std::lock_guard<std::mutex> lk(mu); return rc != 0; }

	void writer()
	{
		for (;;) {
			Item it;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_item.wait(lk, [&] { return rc || ready.count(written) || (workers_left == 0 && ready.empty()); });
				if (rc || !ready.count(written)) return;
				it = ready[written];
				ready.erase(written);
			}
			const double t0 = now_s();
			int64_t o = 0;
			while (fd >= 0 && o < it.len) {
				const ssize_t w = write(fd, it.p + o, (size_t)(it.len - o > (1ll << 30) ? (1ll << 30) : it.len - o));
				if (w < 0) { if (errno == EINTR) continue; fprintf(stderr, "[bwahip] writing the SAM text failed: %s\n", strerror(errno)); fail(BWAHIP_EIO); return; }
				o += w;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				++written; sam_bytes += it.len; t_last_write = now_s(); write_s += t_last_write - t0;
			}
			cv_done.notify_all();
		}
	}
};

// Closing the reader (joining its threads, unmapping gigabytes of input: tens of milliseconds of page-table work) is nobody's critical path:
// it runs on a thread of its own, joined by the next run or when the library is unloaded.  (Putting it off by 200 ms, so that a run that
// follows at once does not open its files beside it -- 20-60 ms instead of 0.5 -- only moved the cost into that run: measured, no gain.)
struct Reaper {
	std::mutex mu; std::thread t;
	void close_later(bwahip_fastq *rd) { std::lock_guard<std::mutex> lk(mu); if (t.joinable()) t.join(); t = std::thread([rd] { bwahip_fastq_close(rd); }); }
	~Reaper() { if (t.joinable()) t.join(); }
};
Reaper g_reaper;

} // namespace

extern "C" int bwahip_stream_run(bwahip_ctx *const *ctxs, int n_ctx, const bwahip_opt_t *opt, const bwahip_pestat_t *pes0,
                                 const char *fq1, const char *fq2, int out_fd, bwahip_stream_t *st)
{
	if (!ctxs || n_ctx < 1 || n_ctx > 256 || !opt || !fq1 || !st) return BWAHIP_EINVAL;
	for (int i = 0; i < n_ctx; ++i) if (!ctxs[i]) return BWAHIP_EINVAL;
	// actual_chunk_size (fastmap.c:304): -K when given, else chunk_size * n_threads
	const int64_t chunk = st->chunk_bases > 0 ? st->chunk_bases : (int64_t)opt->chunk_size * (opt->n_threads > 0 ? opt->n_threads : 1);
	bwahip_opt_t o = *opt;
	if (fq2) o.flag |= BWAHIP_F_PE;
	o.n_threads = opt->n_threads / n_ctx > 1 ? opt->n_threads / n_ctx : 1;   // opt->n_threads is the host-thread budget of the whole run
	Driver d;
	d.fd = out_fd; d.max_reads = st->max_reads;
	const double t_call = now_s();
	int rc = bwahip_fastq_open_mt(fq1, fq2, st->reader_threads, &d.rd);
	if (rc) return rc;
	st->n_reads = st->n_batches = st->sam_bytes = 0; st->seconds = st->reader_wait_s = st->write_s = st->gpu_busy_s = 0;
	const double t_start = now_s();
	d.workers_left = n_ctx;
	d.t_last_write = t_start;
	std::vector<double> wait_s(n_ctx, 0.), busy_s(n_ctx, 0.);
	const int keep_comments = st->keep_comments;
	// Every context has a runner (bwahip_process_seqs_text, batch after batch) and a fetcher that takes the runner's NEXT batch from the
	// reader meanwhile -- linking a million records into a bseq1_t array, and waiting for the parser, are host work that would otherwise
	// sit between two batches of the context.
	struct Slot {
		std::mutex mu; std::condition_variable cv;
		bool full = false, eof = false;
		bwahip_fastq_batch *b = nullptr; bwahip_seq_t *seqs = nullptr; int n = 0; int64_t seq_no = 0, np0 = 0;
	};
	std::vector<Slot> slots(n_ctx);
	auto fetcher = [&](int w) {
		Slot &sl = slots[w];
		for (;;) {
			{ std::unique_lock<std::mutex> lk(sl.mu); sl.cv.wait(lk, [&] { return !sl.full; }); }
			bwahip_fastq_batch *b = nullptr; bwahip_seq_t *seqs = nullptr; int n = 0;
			int64_t seq_no = 0, np0 = 0;
			bool eof = d.failed();
			const double t0 = now_s();
			if (!eof) {
				std::lock_guard<std::mutex> lk(d.mu_read);
				if (d.eof || (d.max_reads > 0 && d.n_processed >= d.max_reads)) { d.eof = true; eof = true; }
				else {
					const int r = bwahip_fastq_next_batch(d.rd, chunk, keep_comments, &b, &seqs, &n);
					if (r) { d.eof = true; eof = true; d.fail(r); }
					else if (n == 0) { d.eof = true; eof = true; }
					else { seq_no = d.next_seq++; np0 = d.n_processed; d.n_processed += n; }
				}
			}
			wait_s[w] += now_s() - t0;
			{ std::lock_guard<std::mutex> lk(sl.mu); sl.b = b; sl.seqs = seqs; sl.n = n; sl.seq_no = seq_no; sl.np0 = np0; sl.eof = eof; sl.full = true; }
			sl.cv.notify_all();
			if (eof) return;
		}
	};
	auto worker = [&](int w) {
		Slot &sl = slots[w];
		std::thread ft(fetcher, w);
		int64_t mine[2] = { -1, -1 };                             // the batches whose text sits in this context's two buffers
		for (int k = 0;; ++k) {
			bwahip_fastq_batch *b; bwahip_seq_t *seqs; int n; int64_t seq_no, np0; bool eof;
			{
				std::unique_lock<std::mutex> lk(sl.mu);
				sl.cv.wait(lk, [&] { return sl.full; });
				b = sl.b; seqs = sl.seqs; n = sl.n; seq_no = sl.seq_no; np0 = sl.np0; eof = sl.eof;
				sl.full = false;
			}
			sl.cv.notify_all();
			if (eof) break;
			bool stop = d.failed();
			// this call overwrites the buffer of this context's last-but-one batch: that one must be on the descriptor
			if (!stop && mine[k & 1] >= 0) { std::unique_lock<std::mutex> lk(d.mu); d.cv_done.wait(lk, [&] { return d.rc || d.written > mine[k & 1]; }); stop = d.rc != 0; }
			if (stop) { bwahip_fastq_batch_release(b); continue; }   // (drain what the fetcher still delivers; it stops at the failure flag)
			const double t1 = now_s();
			const char *sam = nullptr; int64_t len = 0;
			const int r = bwahip_process_seqs_text(ctxs[w], &o, np0, n, seqs, pes0, &sam, &len, nullptr);
			bwahip_fastq_batch_release(b);                          // names, bases and qualities were staged inside the call
			busy_s[w] += now_s() - t1;
			if (r) { d.fail(r); continue; }
			mine[k & 1] = seq_no;
			{ std::lock_guard<std::mutex> lk(d.mu); d.ready[seq_no] = { sam, len }; }
			d.cv_item.notify_all();
		}
		ft.join();
		// the buffers must outlive their write
		{ std::unique_lock<std::mutex> lk(d.mu); d.cv_done.wait(lk, [&] { return d.rc || (d.written > mine[0] && d.written > mine[1]); }); --d.workers_left; }
		d.cv_item.notify_all();
	};
	std::thread wr([&] { d.writer(); });
	std::vector<std::thread> th;
	for (int w = 0; w < n_ctx; ++w) th.emplace_back(worker, w);
	for (auto &t : th) t.join();
	wr.join();
	const double t_joined = now_s();
	g_reaper.close_later(d.rd);
	if (getenv("BWAHIP_STREAM_LOG"))
		fprintf(stderr, "[bwahip] stream: open %.1f ms, first batch in -> last SAM byte out %.1f ms, joining the threads %.1f ms, handing the reader to the closer %.1f ms\n",
		        (t_start - t_call) * 1e3, (d.t_last_write - t_start) * 1e3, (t_joined - d.t_last_write) * 1e3, (now_s() - t_joined) * 1e3);
	st->n_reads = d.n_processed; st->n_batches = d.next_seq; st->sam_bytes = d.sam_bytes;
	st->seconds = d.t_last_write - t_start; st->write_s = d.write_s;
	for (int w = 0; w < n_ctx; ++w) { st->reader_wait_s += wait_s[w]; st->gpu_busy_s += busy_s[w]; }
	return d.rc;
}
