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


# ---------------------------------------------------------------- the parallel pipeline (chunks, stitched boundaries, fallback, BGZF, errors)
def _big_records(n, seed, odd_at=()):
    """n four-line records of random length whose quality lines often start with '@' or '+' (the characters a record-boundary
    search must not fall for); records listed in odd_at are written over several lines (the strict parser has to hand over)."""
    import random
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        l = rnd.randint(1, 260)
        s = "".join(rnd.choice("ACGTN") for _ in range(l))
        q = rnd.choice("@+I#>") + "".join(rnd.choice("@+#5I") for _ in range(l - 1))
        cm = rnd.choice(["", " 1:N:0", "\tBC:Z:AC GT"])
        if i in odd_at and l > 3:
            k = l // 2
            out.append(f"@r{i}/1{cm}\n{s[:k]}\n{s[k:]}\n+r{i}\n{q[:k]}\n{q[k:]}\n")
        else:
            out.append(f"@r{i}/1{cm}\n{s}\n+\n{q}\n")
    return "".join(out).encode()


def _bgzf(data, block=60000):
    """bgzip's container: independent gzip members with the BC extra field, and the empty EOF block."""
    import struct
    import zlib
    out = []
    for o in list(range(0, len(data), block)) + [None]:
        raw = data[o:o + block] if o is not None else b""
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body + struct.pack("<II", zlib.crc32(raw), len(raw)))
    return b"".join(out)


@pytest.mark.parametrize("odd", [(), (20000,), (0,), (39999,)])
def test_parallel_parse_equals_sequential_across_chunks(built, tmp_path, monkeypatch, odd):
    """A 14 MB file (several 4 MB parse chunks per worker): the chunk-parallel parse with stitched boundaries gives the records of
    the sequential kseq_read restatement -- also when a multi-line record in the middle, at the start or at the end forces the
    hand-over -- and of the reference's bseq_read when oracle/_ref is built."""
    import gzip
    data = _big_records(40000, 7, odd)
    p = str(tmp_path / "big.fq")
    open(p, "wb").write(data)
    monkeypatch.setenv("BWAHIP_READER_THREADS", "6")
    got = dump(p, None, 1000000)
    assert got.count(b"\n") - got.count(b"#batch") == 40000
    gz = str(tmp_path / "big.fq.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(data)
    assert dump(gz, None, 1000000) == got                                       # inflate thread + parse workers
    bg = str(tmp_path / "big.bgz.fq.gz")
    open(bg, "wb").write(_bgzf(data))
    assert dump(bg, None, 1000000) == got                                       # BGZF members inflated by the workers
    open(bg, "wb").write(_bgzf(data[:5000000]) + gzip.compress(data[5000000:], 1))   # a plain gzip member after BGZF blocks
    assert dump(bg, None, 1000000) == got
    monkeypatch.setenv("BWAHIP_READER_THREADS", "1")
    monkeypatch.setenv("BWAHIP_READER_NO_MMAP", "1")
    assert dump(p, None, 1000000) == got
    if common.have_ref():
        want = subprocess.run([common.BWAREF, "readfq", "1000000", p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert got == want


def test_no_final_newline_and_tiny_files(built, tmp_path):
    for data, n in ((b"@a\nACGT\n+\nIIII", 1), (b"@a\nACGT\n+\nIIII\n@b\nAC\n+\nII\n", 2), (b"", 0), (b"\n\n", 0), (b">x\nACGT\n>y\nAC\n", 2)):
        p = str(tmp_path / "t.fq")
        open(p, "wb").write(data)
        got = dump(p, None, 100)
        assert got.count(b"\n") - got.count(b"#batch") == n
        if common.have_ref():
            assert got == subprocess.run([common.BWAREF, "readfq", "100", p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout


@pytest.mark.parametrize("damage", ["truncated", "flipped", "bgzf_flipped"])
def test_damaged_gzip_is_an_error_not_a_short_batch(built, tmp_path, damage):
    """err_gzread (utils.c:142) aborts the reference on a gzread error: a truncated or corrupt .gz must not end in a shortened SAM with rc 0."""
    import gzip
    data = _big_records(30000, 3)
    z = bytearray(_bgzf(data) if damage.startswith("bgzf") else gzip.compress(data, 4))
    if damage == "truncated":
        z = z[:len(z) // 2]
    else:
        z[len(z) // 2] ^= 0x55
    p = str(tmp_path / "bad.fq.gz")
    open(p, "wb").write(bytes(z))
    with pytest.raises(bw.BwahipError):
        with bw.FastqReader(p) as r:
            while True:
                _, n = r.next(100000000)
                if n == 0:
                    break


def test_owned_batches_stay_valid_side_by_side(built, tmp_path):
    """bwahip_fastq_next_batch: two batches alive at once (what two contexts in flight need), released in any order, also after close."""
    data = _big_records(30000, 11)
    p = str(tmp_path / "o.fq")
    open(p, "wb").write(data)
    want = dump(p, None, 500000).split(b"\n")
    want = [l for l in want if l and not l.startswith(b"#batch")]
    rd = bw.FastqReader(p, threads=3)
    batches = []
    while True:
        h, arr, n = rd.next_batch(500000, True)
        if n == 0:
            break
        batches.append((h, arr, n))
    rd.close()                                                      # the batches outlive the reader
    assert len(batches) > 4
    i = 0
    for h, arr, n in reversed(batches):
        pass
    for h, arr, n in batches:
        for k in range(n):
            s = arr[k]
            line = b"\t".join([s.name, s.comment if s.comment is not None else b"*", bytes(s.seq[:s.l_seq]), s.qual if s.qual is not None else b"*"])
            assert line == want[i]
            i += 1
    assert i == len(want)
    for h, _, _ in batches[::2] + batches[1::2]:
        bw.FastqReader.release_batch(h)
