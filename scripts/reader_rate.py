#!/usr/bin/env python3
"""Throughput of the product's FASTQ reader (csrc/fastq_reader.cpp) alone: 2 x 1 M reads of 150 bp, plain and gzip.
   python scripts/reader_rate.py"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
bw = entry.load_bwahip()
d = "/dev/shm/reader_rate" if os.path.isdir("/dev/shm") else "/tmp/reader_rate"
os.makedirs(d, exist_ok=True)
n = 1000000
rng = np.random.default_rng(1)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
for m in (1, 2):
    with open(f"{d}/r_{m}.fq", "wb") as f:
        for i in range(n):
            f.write(b"@read%09d/%d\n" % (i, m) + seq[i].tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
    with open(f"{d}/r_{m}.fq", "rb") as f, gzip.open(f"{d}/r_{m}.fq.gz", "wb", compresslevel=4) as g:
        g.write(f.read())
for files in ((f"{d}/r_1.fq", f"{d}/r_2.fq"), (f"{d}/r_1.fq.gz", f"{d}/r_2.fq.gz")):
    t0 = time.time(); tot = 0
    with bw.FastqReader(*files) as rd:
        while True:
            arr, k = rd.next(150000000)
            if k == 0: break
            tot += k
    dt = time.time() - t0
    print(f"{os.path.basename(files[0])} + mate: {tot} reads in {dt:.2f} s = {tot / dt / 1e6:.2f} M reads/s")
