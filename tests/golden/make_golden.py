#!/usr/bin/env python3
"""Generate the committed golden vectors with the REFERENCE ITSELF (oracle/_ref/bwaref, compiled from the
reference's own sources by oracle/Makefile).  Runs only where /root/reference exists (the build container);
the outputs below are data (inputs + expected outputs), never reference source.

  tests/golden/g60k.fa.gz          60 kb three-contig genome with repeats/tandems/N holes (simgen seed 7)
  tests/golden/g60k.alt            ALT contig list (ctg3 is a diverged copy of part of ctg1)
  tests/golden/index.sha256        sha256 of the five index files `bwaref index` writes for it
  tests/golden/se.fq.gz, pe_[12].fq.gz   read sets (mixed error profiles, N's, chimeras, short reads)
  tests/golden/se.sam.gz, se_all.sam.gz, pe.sam.gz   SAM bodies from the reference's mem_process_seqs
  tests/golden/se.stages.npz       per-read stage dump (intervals, chains, filtered chains, regions)
  tests/golden/kat_fm.npz, kat_ksw.npz   known answers for Occ/SA/extend and ksw_extend2/global2/align2
  tests/golden/kat_ksw_align.npz   ksw_align2 under other matrices / gap costs / xtra (word kernel as mem_seed_sw uses it)
  tests/golden/opt_<set>.sam.gz    SAM bodies of se.fq / pe_[12].fq under non-default options (common.GOLDEN_OPTION_SETS)

`make_golden.py extras` writes only the last two groups (round 2 additions); the older files are left alone.
"""
import gzip
import hashlib
import os
import subprocess
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from common import bw  # noqa: E402

REF = common.BWAREF
TMP = "/tmp/bwahip_golden"


def sh(*cmd, **kw):
    return subprocess.run(list(cmd), check=True, **kw)


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        g.write(f.read())


def records_to_npz(path, dst):
    words = np.fromfile(path, dtype=np.int64)
    np.savez_compressed(dst, words=words)


def main():
    assert os.path.exists("/root/reference/bwamem.c") and os.access(REF, os.X_OK), "needs the reference build (make -C oracle ref)"
    os.makedirs(TMP, exist_ok=True)
    fa = f"{TMP}/g60k.fa"
    # genome: two simgen contigs + an ALT contig that is a 2 % diverged copy of ctg1[5000:9000]
    bw.make_genome(fa, 7, [40000, 16000], repeats=True)
    seq = "".join(l.strip() for l in open(fa).read().split(">")[1].split("\n")[1:])
    rng = np.random.default_rng(3)
    alt = list(seq[5000:9000])
    for i in rng.choice(4000, 80, replace=False):
        alt[i] = "ACGT"[("ACGT".index(alt[i]) + 1 + int(rng.integers(0, 3))) % 4] if alt[i] in "ACGT" else alt[i]
    with open(fa, "a") as f:
        f.write(">ctg3 alt_copy\n")
        a = "".join(alt)
        for i in range(0, len(a), 60):
            f.write(a[i:i + 60] + "\n")
    with open(f"{TMP}/g60k.alt", "w") as f:
        f.write("ctg3\n")
    sh(REF, "index", fa, f"{TMP}/g60k")
    digest = {}
    for e in ("pac", "ann", "amb", "bwt", "sa"):
        digest[e] = hashlib.sha256(open(f"{TMP}/g60k.{e}", "rb").read()).hexdigest()
    with open(f"{HERE}/index.sha256", "w") as f:
        for e, h in digest.items():
            f.write(f"{h}  g60k.{e}\n")
    # reads: mix of profiles, concatenated into one SE file
    parts = []
    for k, (n, ln, sub, indel, nn, chim) in enumerate([(120, 150, 10000, 2000, 500, 30000), (80, 100, 20000, 0, 0, 0),
                                                       (60, 250, 50000, 3000, 500, 30000), (30, 30, 20000, 0, 20000, 0),
                                                       (10, 17, 0, 0, 0, 0)]):
        p = f"{TMP}/p{k}.fq"
        bw.make_reads(fa, p, None, n, ln, sub, indel, nn, 300 + k, chim)
        parts.append(open(p).read().replace("@r", f"@s{k}_"))
    open(f"{TMP}/se.fq", "w").write("".join(parts))
    bw.make_reads(fa, f"{TMP}/pe_1.fq", f"{TMP}/pe_2.fq", 400, 100, 30000, 3000, 1000, 310)
    with open(f"{TMP}/se.sam", "wb") as f:
        sh(REF, "mem", f"{TMP}/g60k", f"{TMP}/se.fq", stdout=f, stderr=subprocess.DEVNULL)
    with open(f"{TMP}/se_all.sam", "wb") as f:
        sh(REF, "mem", "-a", f"{TMP}/g60k", f"{TMP}/se.fq", stdout=f, stderr=subprocess.DEVNULL)
    with open(f"{TMP}/pe.sam", "wb") as f:
        sh(REF, "mem", f"{TMP}/g60k", f"{TMP}/pe_1.fq", f"{TMP}/pe_2.fq", stdout=f, stderr=subprocess.DEVNULL)
    sh(REF, "stages", f"{TMP}/g60k", f"{TMP}/se.fq", f"{TMP}/se.stages.bin")
    sh(REF, "katfm", f"{TMP}/g60k", f"{TMP}/kat_fm.bin", "150", "5")
    sh(REF, "katksw", f"{TMP}/kat_ksw.bin", "120", "7")
    gz(fa, f"{HERE}/g60k.fa.gz")
    sh("cp", f"{TMP}/g60k.alt", f"{HERE}/g60k.alt")
    for n in ("se.fq", "pe_1.fq", "pe_2.fq", "se.sam", "se_all.sam", "pe.sam"):
        gz(f"{TMP}/{n}", f"{HERE}/{n}.gz")
    records_to_npz(f"{TMP}/se.stages.bin", f"{HERE}/se.stages.npz")
    records_to_npz(f"{TMP}/kat_fm.bin", f"{HERE}/kat_fm.npz")
    records_to_npz(f"{TMP}/kat_ksw.bin", f"{HERE}/kat_ksw.npz")
    print("golden vectors written to", HERE)
    sh("ls", "-la", HERE)


def extras():
    """Round-2 additions, generated from the committed inputs (g60k.fa.gz, se.fq.gz, pe_[12].fq.gz)."""
    assert os.path.exists("/root/reference/bwamem.c") and os.access(REF, os.X_OK), "needs the reference build (make -C oracle ref)"
    os.makedirs(TMP, exist_ok=True)
    for n in ("g60k.fa", "se.fq", "pe_1.fq", "pe_2.fq"):
        open(f"{TMP}/{n}", "wb").write(gzip.open(f"{HERE}/{n}.gz").read())
    sh("cp", f"{HERE}/g60k.alt", f"{TMP}/g60k.alt")
    sh(REF, "index", f"{TMP}/g60k.fa", f"{TMP}/g60k")
    sh(REF, "katalign", f"{TMP}/kat_ksw_align.bin", "400", "11")
    records_to_npz(f"{TMP}/kat_ksw_align.bin", f"{HERE}/kat_ksw_align.npz")
    # long reads (600-700 bases, 4 % substitutions, indels, chimeras): the seed SW filter drops seeds only there
    parts = []
    for k, ln in enumerate((600, 650, 700)):
        bw.make_reads(f"{TMP}/g60k.fa", f"{TMP}/l{k}.fq", None, 25, ln, 40000, 3000, 500, 320 + k, 50000)
        parts.append(open(f"{TMP}/l{k}.fq").read().replace("@r", f"@l{k}_"))
    open(f"{TMP}/long.fq", "w").write("".join(parts))
    gz(f"{TMP}/long.fq", f"{HERE}/long.fq.gz")
    for name in common.GOLDEN_OPTION_SETS:
        flags = common.option_flags(name)
        fqs = [f"{TMP}/pe_1.fq", f"{TMP}/pe_2.fq"] if name in common.PE_OPTION_SETS else [f"{TMP}/long.fq"] if name in common.LONG_OPTION_SETS else [f"{TMP}/se.fq"]
        with open(f"{TMP}/opt.sam", "wb") as f:
            sh(REF, "mem", *flags, f"{TMP}/g60k", *fqs, stdout=f, stderr=subprocess.DEVNULL)
        gz(f"{TMP}/opt.sam", f"{HERE}/opt_{name}.sam.gz")
    sh("ls", "-la", HERE)


def fastq_cases():
    """tests/golden/fastq: hand-made FASTA/FASTQ inputs and the batches the REFERENCE's bseq_read / kseq_read cut from them
    (`bwaref readfq`, oracle/ref_driver.c) -- the vectors of the product's reader (csrc/fastq_reader.cpp).  Deterministic: running it again
    reproduces the committed files byte for byte."""
    import random
    assert os.path.exists("/root/reference/bwamem.c") and os.access(REF, os.X_OK), "needs the reference build (make -C oracle ref)"
    d = os.path.join(HERE, "fastq")
    os.makedirs(d, exist_ok=True)
    a = (b"@r1/1 first comment  two spaces\nACGTNacgtn\n+\nIIIIIIIIII\n"
         b"@r2\tBC:Z:ACGT\tXY:i:3\r\nACGT\r\nAC\r\n+r2 again\r\nII\r\nIIII\r\n"          # CRLF, two-line sequence and quality
         b"\n\n@r3/2\nACGTACGTAC\n+\n@IIIIIIII@\n"                                          # quality starting with '@'; empty lines between records
         b"@r4 \nAC\n+\nII\n"                                                                   # a single blank after the name: empty comment
         b">fa1 a fasta record\nACGT\nACGTAC\n\nGG\n"                                          # FASTA, three sequence lines and an empty one
         b"@r5/x\n\n+\n\n"                                                                      # empty sequence; '/x' is not a read number
         b"@r6/12\nACGTA\n+\nIIIII\n"                                                          # only the last two characters are looked at
         b"@last\nACGTACGT\n+\nIIII")                                                          # truncated quality: the record is dropped
    open(f"{d}/cases_1.fq", "wb").write(a)
    b = b"".join(b"@m%d/2 mate\nTTTTGGGGCC%s\n+\n%s\n" % (i, b"A" * i, b"J" * (10 + i)) for i in range(6))      # fewer mates than reads in cases_1.fq
    open(f"{d}/cases_2.fq", "wb").write(b)
    random.seed(5)

    def rec(i, mate):
        l = random.randint(30, 60)
        s = "".join(random.choice("ACGT") for _ in range(l))
        return f"@p{i}/{mate} c{i}\n{s}\n+\n{'F' * l}\n".encode()
    open(f"{d}/pairs_1.fq", "wb").write(b"".join(rec(i, 1) for i in range(400)))
    with gzip.GzipFile(f"{d}/pairs_2.fq.gz", "wb", mtime=0) as g:
        g.write(b"".join(rec(i, 2) for i in range(400)))
    for out, chunk, files in (("cases_1.se.txt", 100000, ["cases_1.fq"]), ("cases.pe.txt", 100000, ["cases_1.fq", "cases_2.fq"]),
                              ("pairs.pe3000.txt", 3000, ["pairs_1.fq", "pairs_2.fq.gz"]), ("pairs.se777.txt", 777, ["pairs_1.fq"])):
        with open(f"{d}/{out}", "wb") as f:
            sh(REF, "readfq", str(chunk), *[f"{d}/{x}" for x in files], stdout=f, stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    {"extras": extras, "fastq": fastq_cases}.get(sys.argv[1] if len(sys.argv) > 1 else "", main)()
