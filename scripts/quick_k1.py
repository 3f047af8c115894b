"""Interim measurement of the seeding kernels on a CPU-built index (development aid, not the bench)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from common import bw

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
d = "/tmp/qk1"; os.makedirs(d, exist_ok=True)
fa, prefix, fq = f"{d}/g.fa", f"{d}/g", f"{d}/r.fq"
t0 = time.time()
bw.make_genome(fa, 38, [mbp * 1000000], repeats=True)
bw.make_index(fa, prefix)
print(f"genome+index {mbp} Mbp: {time.time()-t0:.1f}s", flush=True)
bw.make_reads(fa, fq, None, n_reads, 150, 10000, 0, 0, 102, 0)
_, seqs, _ = bw.read_fastq(fq)
codes, off = bw.pack_reads(seqs)
ctx = bw.Context(prefix)
ctx.batch_upload(codes, off)
for it in range(4):
    t0 = time.time()
    ms = ctx.batch_run()
    wall = time.time() - t0
    cn = ctx.counters()
    k1 = ms["k_smem"]
    blocks = cn["blocks"]
    print(f"iter {it}: wall {wall*1e3:.1f} ms kernels {ms} counters {cn}")
    print(f"   k_smem: {n_reads/k1*1e3/1e6:.2f} M reads/s, {cn['extend']/n_reads:.1f} extends/read, {blocks*64/k1*1e3/1e9:.1f} GB/s algorithmic; "
          f"k_seeds: {(cn['lf']*64+cn['sa']*8)/max(ms['k_seeds'],1e-6)*1e3/1e9:.1f} GB/s", flush=True)
