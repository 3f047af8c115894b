#!/usr/bin/env python3
"""Host-buffer boundary rate: bwahip_process_seqs (host bseq1_t in, SAM text out; GPU hot path + host finalisation on
opt.n_threads threads, PCIe both ways) on a sample of the bench workload.  Reported in profiles/, never as bench `value`.
usage: python scripts/e2e_rate.py [genome_mbp] [reads] [threads]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
bw = entry.load_bwahip(); bw.lib()
import tools_py as tp

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
thr = int(sys.argv[3]) if len(sys.argv) > 3 else min(64, os.cpu_count() or 1)
work = "/dev/shm/bwahip_bench"; os.makedirs(work, exist_ok=True)
prefix = os.path.join(work, f"g{mbp}")
lens = tp.contig_lengths(mbp * 1000000)
genome = tp.make_genome(38, lens, repeats=True)
if not os.path.exists(prefix + ".sa"):
    tp.write_fasta(prefix + ".fa", genome, lens); bw.make_index(prefix + ".fa", prefix)
reads = tp.make_reads(genome, lens, n, 150, sub_ppm=10000, seed=102)
acgt = np.frombuffer(b"ACGTN", dtype=np.uint8)
seqs = [bytes(r) for r in reads] if reads.dtype == np.uint8 and reads.max() > 4 else [acgt[r].tobytes() for r in reads]
names = [b"r%d" % i for i in range(n)]
opt = bw.default_opt(); opt.n_threads = thr
with bw.Context(prefix, 0) as ctx:
    arr = (bw.Seq * n)(); keep = []
    def fill():
        keep.clear()
        for i in range(n):
            sb = C.create_string_buffer(seqs[i], len(seqs[i]) + 1); keep.append(sb)
            arr[i].l_seq, arr[i].id, arr[i].name, arr[i].comment = len(seqs[i]), i, names[i], None
            arr[i].seq, arr[i].qual = C.cast(sb, C.POINTER(C.c_char)), None
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    best = None
    for rep in range(3):
        fill()
        t0 = time.time()
        rc = bw.lib().bwahip_process_seqs(ctx._h, C.byref(opt), 0, n, arr, None)
        dt = time.time() - t0
        assert rc == 0, rc
        sam_bytes = sum(len(C.string_at(arr[i].sam)) for i in range(0, n, 997)) * 997
        for i in range(n):
            libc.free(C.cast(arr[i].sam, C.c_void_p))
        best = dt if best is None or dt < best else best
        print(f"rep {rep}: {n / dt:,.0f} reads/s ({dt * 1e3:.0f} ms for {n} reads, {thr} host threads, ~{sam_bytes / 1e6:.0f} MB of SAM)", flush=True)
print(f"bwahip_process_seqs: {n / best:,.0f} reads/s end to end (host buffers in, SAM text out) on {mbp} Mbp, {thr} host threads")
