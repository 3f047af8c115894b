"""The product's FASTA/FASTQ reader (csrc/fastq_reader.cpp, bwahip_fastq_*) against the batches the REFERENCE's bseq_read /
kseq_read cut from the same files (tests/golden/fastq/*.txt, written by `oracle/_ref/bwaref readfq`: hand-made inputs with
comments, CRLF line ends, multi-line records, FASTA, '@' in qualities, empty sequences, a truncated last record, a mates' file
with fewer reads, a gzip file; and the -K chunking rule).  CPU only: the reader needs no GPU."""
import os
import subprocess
import pytest
import common
from common import bw

F = os.path.join(common.GOLDEN, "fastq")


def dump(path1, path2, chunk, keep_comments=True):
    out = []
    with bw.FastqReader(path1, path2) as r:
        while True:
            arr, n = r.next(chunk, keep_comments)
            if n == 0:
                break
            out.append(b"#batch %d\n" % n)
            for i in range(n):
                s = arr[i]
                assert s.id == i and not s.sam
                seq = bytes(s.seq[:s.l_seq])
                out.append(b"\t".join([s.name, s.comment if s.comment is not None else b"*", seq, s.qual if s.qual is not None else b"*"]) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("want,f1,f2,chunk", [("cases_1.se.txt", "cases_1.fq", None, 100000), ("cases.pe.txt", "cases_1.fq", "cases_2.fq", 100000),
                                              ("pairs.pe3000.txt", "pairs_1.fq", "pairs_2.fq.gz", 3000), ("pairs.se777.txt", "pairs_1.fq", None, 777)])
def test_reader_equals_reference_bseq_read(built, want, f1, f2, chunk):
    got = dump(os.path.join(F, f1), os.path.join(F, f2) if f2 else None, chunk)
    assert got == open(os.path.join(F, want), "rb").read()


def test_comments_dropped_without_C(built):
    got = dump(os.path.join(F, "cases_1.fq"), None, 100000, keep_comments=False)
    assert all(l.split(b"\t")[1] == b"*" for l in got.split(b"\n") if l and not l.startswith(b"#"))


@pytest.mark.skipif(not common.have_ref(), reason="oracle/_ref/bwaref not built (no /root/reference here)")
def test_reader_equals_reference_on_a_fresh_file(built, tmp_path):
    """A fresh random file pair through both readers (whenever the reference binary is present)."""
    import random
    random.seed(int.from_bytes(os.urandom(4), "little"))
    def rec(i, mate):
        l = random.randint(0, 90)
        s = "".join(random.choice("ACGTNacgt") for _ in range(l))
        cm = random.choice(["", " x", "\tBC:Z:A C", "  two"])
        if random.random() < .2:                                    # multi-line
            k = l // 2
            return f"@q{i}/{mate}{cm}\n{s[:k]}\n{s[k:]}\n+\n{'E' * k}\n{'E' * (l - k)}\n"
        return f"@q{i}/{mate}{cm}\n{s}\n+\n{'#' * l}\n"
    p1, p2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(p1, "w").write("".join(rec(i, 1) for i in range(3000)))
    open(p2, "w").write("".join(rec(i, 2) for i in range(3000)))
    for chunk in (500, 20000):
        want = subprocess.run([common.BWAREF, "readfq", str(chunk), p1, p2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert dump(p1, p2, chunk) == want
