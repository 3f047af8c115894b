#!/usr/bin/env python3
"""Throughput of the product's FASTQ reader (csrc/fastq_reader.cpp) alone: 2 x N reads of 150 bp as plain files, gzip (one member:
inflate on its own thread per file, parse workers behind it) and BGZF (bgzip blocks inflated by the workers in parallel).
   python scripts/reader_rate.py [reads_per_file] [threads]"""
import gzip, os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
bw = entry.load_bwahip()
d = "/dev/shm/reader_rate" if os.path.isdir("/dev/shm") else "/tmp/reader_rate"
os.makedirs(d, exist_ok=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def bgzf(data, block=65280):
    out = []
    for o in list(range(0, len(data), block)) + [None]:
        raw = data[o:o + block] if o is not None else b""
        co = zlib.compressobj(4, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(raw), len(raw)))
    return b"".join(out)


rng = np.random.default_rng(1)
for m in (1, 2):
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
    rec = np.empty((n, 17 + 151 + 2 + 151), dtype=np.uint8)
    rec[:, :17] = np.array([b"@read%09d/%d\n" % (i, m) for i in range(n)], dtype="S17").view(np.uint8).reshape(n, 17)
    rec[:, 17:167] = seq; rec[:, 167] = 10; rec[:, 168] = ord("+"); rec[:, 169] = 10; rec[:, 170:320] = ord("I"); rec[:, 320] = 10
    rec.tofile(f"{d}/r_{m}.fq")
    k = min(n, 1000000)                                            # the compressed variants: 1 M reads per file is enough for a rate
    part = rec[:k].tobytes()
    with gzip.open(f"{d}/r_{m}.fq.gz", "wb", compresslevel=4) as g:
        g.write(part)
    open(f"{d}/r_{m}.bgz.fq.gz", "wb").write(bgzf(part))
for files in ((f"{d}/r_1.fq", f"{d}/r_2.fq"), (f"{d}/r_1.fq.gz", f"{d}/r_2.fq.gz"), (f"{d}/r_1.bgz.fq.gz", f"{d}/r_2.bgz.fq.gz")):
    for rep in range(2):
        t0 = time.time(); tot = 0
        with bw.FastqReader(*files, threads=threads) as rd:
            while True:
                arr, k = rd.next(150000000)
                if k == 0: break
                tot += k
        dt = time.time() - t0
    print(f"{os.path.basename(files[0])} + mate: {tot} reads in {dt:.2f} s = {tot / dt / 1e6:.2f} M reads/s ({threads or 'default'} parse threads, {os.cpu_count()} cpus)", flush=True)
