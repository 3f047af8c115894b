"""End-to-end drop-in check: bwahip_process_seqs (hot path on the GPU + host finalisation) must produce
byte-identical SAM to the CPU path (oracle == reference's mem_process_seqs) on the same reads: every column,
i.e. POS, CIGAR, FLAG, MAPQ and all tags."""
import os
import subprocess
import numpy as np
import pytest
import common
from common import bw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(small_index):
    c = bw.Context(small_index["prefix"])
    yield c
    c.close()


def _oracle_sam(prefix, fqs, extra=()):
    out = subprocess.run([common.ORACLE, "mem", "-t", "8", *extra, prefix, *fqs], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    return out.stdout


@pytest.mark.parametrize("name,n,length,sub,indel,nn,seed,chim", [
    ("se150", 4000, 150, 10000, 2000, 500, 201, 20000),
    ("se250", 1500, 250, 50000, 3000, 500, 205, 30000),
    ("se100", 3000, 100, 20000, 1000, 0, 202, 0),
])
def test_se_sam_identical(ctx, small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim):
    fq = str(tmp_path / f"{name}.fq")
    bw.make_reads(small_index["fa"], fq, None, n, length, sub, indel, nn, seed, chim)
    names, seqs, quals = bw.read_fastq(fq)
    want = _oracle_sam(small_index["prefix"], [fq])
    opt = bw.default_opt()
    opt.n_threads = 8
    got = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    assert got == want, _first_diff(got, want)


def test_se_sam_identical_across_batches(ctx, small_index, tmp_path):
    """n_processed feeds the tie-breaking hash (bwamem.c:534,1204): split batches must reproduce one big batch."""
    fq = str(tmp_path / "b.fq")
    bw.make_reads(small_index["fa"], fq, None, 3000, 150, 10000, 2000, 500, 207, 20000)
    names, seqs, quals = bw.read_fastq(fq)
    want = _oracle_sam(small_index["prefix"], [fq])
    opt = bw.default_opt()
    opt.n_threads = 4
    got = b""
    for b0 in range(0, 3000, 1000):
        got += b"".join(ctx.process_seqs(names[b0:b0+1000], seqs[b0:b0+1000], quals[b0:b0+1000], opt, n_processed=b0))
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("name,n,length,sub,indel,nn,seed", [
    ("pe150", 6000, 150, 10000, 1000, 300, 203),
    ("pe100_noisy", 6000, 100, 60000, 5000, 2000, 204),
    ("pe200_noisy", 3000, 200, 40000, 4000, 1000, 205),       # mates of 161..249 bases: sixteen cells per lane in the byte kernel of the mate rescue
    ("pe260_noisy", 3000, 260, 40000, 4000, 1000, 206),       # 250 bases and more: the word kernel (ksw_i16) does the rescue alignments
])
def test_pe_sam_identical(ctx, small_index, tmp_path, name, n, length, sub, indel, nn, seed):
    fq1, fq2 = str(tmp_path / f"{name}_1.fq"), str(tmp_path / f"{name}_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, n, length, sub, indel, nn, seed)
    n1, s1, q1 = bw.read_fastq(fq1)
    n2, s2, q2 = bw.read_fastq(fq2)
    names = [x for p in zip(n1, n2) for x in p]
    seqs = [x for p in zip(s1, s2) for x in p]
    quals = [x for p in zip(q1, q2) for x in p]
    want = _oracle_sam(small_index["prefix"], [fq1, fq2])
    opt = bw.default_opt()
    opt.n_threads = 8
    opt.flag |= 0x2
    got = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    assert got == want, _first_diff(got, want)


def test_pe_stages_ran_on_the_gpu_and_match_host_finalisation(ctx, small_index, tmp_path):
    """The paired-end path end to end on the GPU (insert-size histogram, mate rescue with the lane-exact striped SW,
    mem_pair, paired SAM): mate-rescue alignments really ran there, the statistics are the reference's, and the SAM is
    the same whether finalisation runs on the GPU or (knobs gpu_final / gpu_pair = 0) on host threads."""
    fq1, fq2 = str(tmp_path / "x_1.fq"), str(tmp_path / "x_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, 5000, 150, 40000, 4000, 2000, 233)
    n1, s1, q1 = bw.read_fastq(fq1)
    n2, s2, q2 = bw.read_fastq(fq2)
    names = [x for p in zip(n1, n2) for x in p]
    seqs = [x for p in zip(s1, s2) for x in p]
    quals = [x for p in zip(q1, q2) for x in p]
    r = subprocess.run([common.ORACLE, "mem", "-t", "8", small_index["prefix"], fq1, fq2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    opt = bw.default_opt()
    opt.n_threads = 8
    opt.flag |= 0x2
    got = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    assert got == r.stdout, _first_diff(got, r.stdout)
    pes, n_sw, n_new = ctx.last_pe_stats()
    assert n_sw > 50 and n_new > 10, (n_sw, n_new)                       # 4 % substitutions: many mates need rescue
    assert pes[1]["failed"] == 0 and 450 < pes[1]["avg"] < 550 and pes[0]["failed"] == pes[2]["failed"] == pes[3]["failed"] == 1
    try:
        ctx.tune(gpu_pair=0)
        host = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    finally:
        ctx.tune(gpu_pair=1)
    assert host == got
    # single-end: GPU finalisation vs host finalisation
    opt.flag &= ~0x2
    a = b"".join(ctx.process_seqs(n1, s1, q1, opt))
    try:
        ctx.tune(gpu_final=0)
        b = b"".join(ctx.process_seqs(n1, s1, q1, opt))
    finally:
        ctx.tune(gpu_final=1)
    assert a == b


def _first_diff(got, want):
    g, w = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(g, w)):
        if a != b:
            return f"SAM differs at line {i} of {len(w)}:\n got  {a[:300]}\n want {b[:300]}"
    return f"SAM line counts differ: {len(g)} vs {len(w)}"


def test_golden_reference_sam_on_gpu(built, tmp_path):
    """HIP path vs SAM bodies the REFERENCE itself produced (tests/golden, generated by oracle/_ref/bwaref):
    SE (with ALT contig, chimeras, N's, reads shorter than a seed) and PE."""
    import gzip
    G = common.GOLDEN
    fa = str(tmp_path / "g60k.fa")
    open(fa, "wb").write(gzip.open(os.path.join(G, "g60k.fa.gz")).read())
    bw.make_index(fa, str(tmp_path / "g60k"))
    open(str(tmp_path / "g60k.alt"), "wb").write(open(os.path.join(G, "g60k.alt"), "rb").read())
    for n in ("se.fq", "pe_1.fq", "pe_2.fq"):
        open(str(tmp_path / n), "wb").write(gzip.open(os.path.join(G, n + ".gz")).read())
    with bw.Context(str(tmp_path / "g60k")) as c:
        names, seqs, quals = bw.read_fastq(str(tmp_path / "se.fq"))
        opt = bw.default_opt()
        opt.n_threads = 4
        got = b"".join(c.process_seqs(names, seqs, quals, opt))
        want = gzip.open(os.path.join(G, "se.sam.gz")).read()
        assert got == want, _first_diff(got, want)
        opt.flag |= 0x8                                                   # -a
        got = b"".join(c.process_seqs(names, seqs, quals, opt))
        want = gzip.open(os.path.join(G, "se_all.sam.gz")).read()
        assert got == want, _first_diff(got, want)
        n1, s1, q1 = bw.read_fastq(str(tmp_path / "pe_1.fq"))
        n2, s2, q2 = bw.read_fastq(str(tmp_path / "pe_2.fq"))
        opt = bw.default_opt()
        opt.n_threads = 4
        opt.flag |= 0x2
        got = b"".join(c.process_seqs([x for p in zip(n1, n2) for x in p], [x for p in zip(s1, s2) for x in p],
                                      [x for p in zip(q1, q2) for x in p], opt))
        want = gzip.open(os.path.join(G, "pe.sam.gz")).read()
        assert got == want, _first_diff(got, want)
        # stage boundaries against the reference's own stage dump
        want_st = common.by_read(bw.parse_records(np.load(os.path.join(G, "se.stages.npz"))["words"]))
        codes, off = bw.pack_reads(seqs)
        got_st = common.by_read(c.run_stages(codes, off, [bw.STAGE_INTV, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS]))
        for st in (bw.STAGE_INTV, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS):
            common.assert_stage_equal(got_st, want_st, st, f"golden stage {st}")


def test_plain_c_caller_reproduces_golden_sam(built, tmp_path):
    """tests/c_abi_driver (plain C; links libbwahip.so and libbwamem_hip.so, no ctypes): bwahip_init on caller-owned
    bwt_t/bntseq_t/pac structs -> bwahip_align_batch -> mem_process_seqs with the reference's signature (bwamem.h:69).
    Its SAM must equal what the REFERENCE produced (tests/golden/se.sam.gz, pe.sam.gz), also when the reads come in
    several batches (n_processed) and with a read group."""
    import gzip
    G = common.GOLDEN
    fa = str(tmp_path / "g60k.fa")
    open(fa, "wb").write(gzip.open(os.path.join(G, "g60k.fa.gz")).read())
    bw.make_index(fa, str(tmp_path / "g60k"))
    open(str(tmp_path / "g60k.alt"), "wb").write(open(os.path.join(G, "g60k.alt"), "rb").read())
    for n in ("se.fq", "pe_1.fq", "pe_2.fq"):
        open(str(tmp_path / n), "wb").write(gzip.open(os.path.join(G, n + ".gz")).read())
    drv = os.path.join(common.ROOT, "tests", "c_abi_driver")
    prefix = str(tmp_path / "g60k")
    run = lambda *a: subprocess.run([drv, *a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    r = run("-t", "4", prefix, str(tmp_path / "se.fq"))
    want = gzip.open(os.path.join(G, "se.sam.gz")).read()
    assert r.stdout == want, _first_diff(r.stdout, want)
    assert r.stderr.decode().startswith("regs 300 ") or b"regs " in r.stderr                   # the bwahip_align_batch pass ran
    r2 = run("-t", "4", "-K", "64", prefix, str(tmp_path / "se.fq"))                            # 64-read batches, true n_processed
    assert r2.stdout == want, _first_diff(r2.stdout, want)
    assert [l for l in r2.stderr.decode().splitlines() if l.startswith("regs ")] == [l for l in r.stderr.decode().splitlines() if l.startswith("regs ")]
    r = run("-t", "4", "-a", prefix, str(tmp_path / "se.fq"))
    want_all = gzip.open(os.path.join(G, "se_all.sam.gz")).read()
    assert r.stdout == want_all, _first_diff(r.stdout, want_all)
    r = run("-t", "4", prefix, str(tmp_path / "pe_1.fq"), str(tmp_path / "pe_2.fq"))
    want_pe = gzip.open(os.path.join(G, "pe.sam.gz")).read()
    assert r.stdout == want_pe, _first_diff(r.stdout, want_pe)
    r = run("-t", "2", "-R", "grp1", prefix, str(tmp_path / "se.fq"))                          # RG:Z: tag after XS (bwamem.c:920)
    exp = subprocess.run([common.ORACLE, "mem", "-t", "2", "-R", "@RG\\tID:grp1\\tSM:x", prefix, str(tmp_path / "se.fq")],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert r.stdout == exp, _first_diff(r.stdout, exp)
    assert b"\tRG:Z:grp1" in r.stdout
    # the loop of INTEGRATION.md 1b from C: the library's reader (one file gzipped), SAM in one piece, batches alternating between a context
    # and its clone; 9000-base batches = 60 reads at a time, cut exactly where the reference's bseq_read cuts them
    import gzip as _gz
    _gz.open(str(tmp_path / "pe_2.fq.gz"), "wb").write(open(str(tmp_path / "pe_2.fq"), "rb").read())
    r = run("-F", "-t", "4", prefix, str(tmp_path / "pe_1.fq"), str(tmp_path / "pe_2.fq.gz"))
    assert r.stdout == want_pe, _first_diff(r.stdout, want_pe)
    r = run("-F", "-t", "4", "-K", "9000", prefix, str(tmp_path / "se.fq"))
    assert r.stdout == want, _first_diff(r.stdout, want)
    assert b"files 300 reads" in r.stderr or b"files " in r.stderr


def test_comments_and_read_group(ctx, small_index, tmp_path):
    """-C (FASTQ comment appended to every record, bwamem.c:955) and -R (RG:Z: tag, bwamem.c:947) through the GPU SAM kernels,
    SE and PE, vs the CPU path given the same flags."""
    fq1, fq2 = str(tmp_path / "c_1.fq"), str(tmp_path / "c_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, 1200, 150, 10000, 2000, 500, 171, 20000)
    comments = []
    for path in (fq1, fq2):                                     # two reads in three carry a comment
        lines = open(path).read().split("\n")
        for i in range(0, len(lines) - 3, 4):
            k = i // 4
            c = f"BC:Z:{'ACGT'[k % 4] * 6}\tXZ:i:{k}" if k % 3 else None
            if path == fq1:
                comments.append(c)
            if c:
                lines[i] += " " + c
        open(path, "w").write("\n".join(lines))
    n1, s1, q1 = bw.read_fastq(fq1)
    n2, s2, q2 = bw.read_fastq(fq2)
    rg = ["-R", "@RG\\tID:grp7\\tSM:sample"]
    body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
    cm = [c.encode() if c else None for c in comments]
    ctx.set_rg_id("grp7")
    try:
        opt = bw.default_opt()
        opt.n_threads = 4
        want = subprocess.run([common.ORACLE, "mem", "-t", "4", "-C", *rg, small_index["prefix"], fq1], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got = b"".join(ctx.process_seqs(n1, s1, q1, opt, comments=cm))
        assert b"RG:Z:grp7" in got and b"BC:Z:CCCCCC" in got
        assert got == body(want)
        opt.flag |= 0x2
        want = subprocess.run([common.ORACLE, "mem", "-t", "4", "-C", *rg, small_index["prefix"], fq1, fq2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        names = [x for p in zip(n1, n2) for x in p]; seqs = [x for p in zip(s1, s2) for x in p]; quals = [x for p in zip(q1, q2) for x in p]
        got = b"".join(ctx.process_seqs(names, seqs, quals, opt, comments=[c for c in cm for _ in (0, 1)]))
        assert got == body(want)
    finally:
        ctx.set_rg_id(None)


def test_files_through_the_product_reader(ctx, small_index, tmp_path):
    """FASTQ files (one of them gzipped) -> bwahip_fastq_* -> bwahip_process_seqs, batch by batch with -K 30000, vs the CPU path
    reading the same files with the same -K: the reader's batches, n_processed and the SAM must all line up (SE and PE)."""
    import gzip
    fq1, fq2 = str(tmp_path / "f_1.fq"), str(tmp_path / "f_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, 1500, 150, 10000, 2000, 500, 181, 20000)
    gz2 = fq2 + ".gz"
    gzip.open(gz2, "wb").write(open(fq2, "rb").read())
    for files in ((fq1,), (fq1, gz2)):
        opt = bw.default_opt()
        opt.n_threads = 4
        if len(files) == 2:
            opt.flag |= 0x2
        want = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", "30000", small_index["prefix"], *files], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got, n_processed, n_batches = [], 0, 0
        with bw.FastqReader(*files) as rd:
            while True:
                arr, n = rd.next(30000)
                if n == 0:
                    break
                text, off = ctx.process_seqs_text_array(arr, n, opt, n_processed=n_processed, want_offsets=True)   # SAM in one piece ...
                assert off[0] == 0 and off[n] == len(text) and all(off[i] <= off[i + 1] for i in range(n))
                assert all(text[off[i]:off[i + 1]].endswith(b"\n") for i in range(0, n, 97))
                assert not arr[0].sam
                got.append(ctx.process_seqs_array(arr, n, opt, n_processed=n_processed))                              # ... and per read: the same bytes
                assert got[-1] == text
                n_processed += n
                n_batches += 1
        assert n_batches >= 4 and n_processed == 750 * len(files)
        body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
        assert b"".join(got) == body(want)


def test_reads_inside_a_large_repeat_family(built, tmp_path):
    """Reads from a 400-bp unit that occurs 700 times (2 % diverged copies) in a 3 Mbp genome: hundreds of chains and regions per read.
    These reads take the heavy-read paths -- k_smem_heavy, k_chain_big / k_chain_flt, k_extend_spec, the bitonic index sorts of k_mark and
    k_pair (lists over 128 regions), mark_core in kept-region order -- which ordinary reads never reach; SAM with -a (every hit printed)
    and without, SE and PE, must equal the CPU path's."""
    import numpy as np
    rng = np.random.default_rng(77)
    L = 3000000
    g = rng.integers(0, 4, L, dtype=np.uint8)
    unit = rng.integers(0, 4, 400, dtype=np.uint8)
    starts = np.sort(rng.choice(np.arange(1000, L - 1400, 1400), 700, replace=False))
    for s in starts:
        c = unit.copy()
        m = rng.random(400) < 0.02
        c[m] = (c[m] + rng.integers(1, 4, int(m.sum()))) % 4
        g[s:s + 400] = c
    fa = str(tmp_path / "rep.fa")
    seq = b"ACGT"
    txt = bytes(np.frombuffer(seq, dtype=np.uint8)[g])
    with open(fa, "wb") as f:
        f.write(b">rep\n")
        for i in range(0, L, 60):
            f.write(txt[i:i + 60] + b"\n")
    prefix = str(tmp_path / "rep")
    bw.make_index(fa, prefix)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    def fq(path, reads):
        with open(path, "wb") as f:
            for i, r in enumerate(reads):
                f.write(b"@h%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    r1, r2 = [], []
    for i in range(60):
        s = int(starts[rng.integers(0, 700)])
        if i % 3 == 0:                                              # fragment inside the repeat unit and its unique flank
            a = s + int(rng.integers(0, 200)); frag = txt[a:a + 420]
        else:                                                       # fragment wholly inside the unit: every copy matches
            a = s + int(rng.integers(0, 60)); frag = txt[a:a + 330]
        r1.append(frag[:150]); r2.append(frag[-150:].translate(comp)[::-1])
    f1, f2 = str(tmp_path / "h_1.fq"), str(tmp_path / "h_2.fq")
    fq(f1, r1); fq(f2, r2)
    body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
    with bw.Context(prefix) as c:
        for extra, flag in (([], 0), (["-a"], 0x8)):
            for files in ((f1,), (f1, f2)):
                opt = bw.default_opt()
                opt.n_threads = 4
                opt.flag |= flag | (0x2 if len(files) == 2 else 0)
                want = subprocess.run([common.ORACLE, "mem", "-t", "4", *extra, prefix, *files], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                with bw.FastqReader(*files) as rd:
                    arr, n = rd.next(10 ** 9)
                    got = c.process_seqs_text_array(arr, n, opt)
                assert got == body(want), f"heavy reads: SAM differs ({extra}, {len(files)} file(s))"
                if extra:
                    assert got.count(b"\n") > 10 * n                                   # -a: every hit above the drop ratio is a record
        cnt = c.counters()
        codes, off = bw.pack_reads(r1)
        regs = common.by_read(c.run_stages(codes, off, [bw.STAGE_REGS]))
        n_regs = max(int(regs[i][bw.STAGE_REGS][0]) for i in range(len(r1)))
    assert cnt["max_chains"] > 128, cnt["max_chains"]
    assert n_regs > 128, n_regs                                  # the lists the bitonic sorts of k_mark / k_pair take
