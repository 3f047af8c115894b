#!/usr/bin/env python3
"""bwahip_stream_run alone on the files a bench.py run left in /dev/shm/bwahip_bench (index g3100, bench_r0_[12].fq): for a kernel trace of the
file-to-file path (rocprofv3 --kernel-trace -- python3 scripts/stream_only.py [contexts] [passes]).  Prints reads/s per pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
bw = entry.load_bwahip()
d = os.environ.get("BWAHIP_BENCH_DIR", "/dev/shm/bwahip_bench")
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prefix = os.path.join(d, os.environ.get("BWAHIP_BENCH_INDEX", "g3100"))
fq1, fq2 = os.path.join(d, "bench_r0_1.fq"), os.path.join(d, "bench_r0_2.fq")
opt = bw.default_opt(); opt.flag |= 2; opt.n_threads = int(os.environ.get("BWAHIP_BENCH_HOST_THREADS", "16"))
with bw.Context(prefix, 0) as c0:
    ctxs = [c0] + [c0.clone() for _ in range(n_ctx - 1)]
    fd = os.open("/dev/null", os.O_WRONLY)
    for p in range(passes):
        t0 = time.time()
        st = bw.stream_run(ctxs, fq1, fq2, fd, opt, chunk_bases=150000000, reader_threads=int(os.environ.get("BWAHIP_BENCH_READER_THREADS", "8")))
        dt = time.time() - t0
        print(f"pass {p}: {st.n_reads} reads in {dt:.3f}s ({st.n_reads / dt / 1e6:.2f} M reads/s); inside driver {st.seconds:.3f}s, workers waiting for the reader {st.reader_wait_s:.3f}s, in process_seqs {st.gpu_busy_s:.3f}s", flush=True)
    os.close(fd)
    for c in ctxs[1:]:
        c.close()
