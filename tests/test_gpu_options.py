"""Option sweep: the HIP path under non-default `bwa mem` options (seeding -k -r -y -c -s, chaining -W -G -N -D -X, extension
-w -d -A -B -O -E -L, output -T -a -M -Y -5 -q -h -Q, PE -m -U -S -P -I) against the CPU path with the same options:
every stage boundary bit-exact and the SAM byte-identical.  -W switches mem_flt_chained_seeds (bwamem.c:605, the seed
SW filter, k_seed_sw on the GPU) on for ordinary reads."""
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


@pytest.fixture(scope="module")
def se_reads(small_index, tmp_path_factory):
    d = tmp_path_factory.mktemp("optsweep")
    fq = str(d / "se.fq")
    parts = []
    for k, (n, ln, sub, indel, nn, chim) in enumerate([(1200, 150, 15000, 3000, 500, 30000), (400, 250, 50000, 4000, 500, 30000), (300, 80, 20000, 2000, 0, 0)]):
        p = str(d / f"p{k}.fq")
        bw.make_reads(small_index["fa"], p, None, n, ln, sub, indel, nn, 700 + k, chim)
        parts.append(open(p).read().replace("@r", f"@s{k}_"))
    open(fq, "w").write("".join(parts))
    f1, f2 = str(d / "pe_1.fq"), str(d / "pe_2.fq")
    bw.make_reads(small_index["fa"], f1, f2, 3000, 150, 20000, 2000, 500, 710)
    return {"fq": fq, "pe": (f1, f2), "dir": str(d)}


def _body(s):
    return b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))


@pytest.mark.parametrize("name", sorted(common.OPTION_SETS))
def test_se_stages_and_sam_under_options(ctx, small_index, se_reads, tmp_path, name):
    flags = common.OPTION_SETS[name]
    opt, _ = common.opt_from_cli(flags)
    opt.n_threads = 8
    fq = se_reads["fq"]
    names, seqs, quals = bw.read_fastq(fq)
    obin = str(tmp_path / "o.bin")
    subprocess.check_call([common.ORACLE, "stages", *flags, small_index["prefix"], fq, obin])
    want = common.by_read(bw.read_record_file(obin))
    codes, off = bw.pack_reads(seqs)
    stages = [bw.STAGE_INTV, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS]
    got = common.by_read(ctx.run_stages(codes, off, stages, opt))
    for st, what in [(bw.STAGE_INTV, "intervals"), (bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS_PRE, "regions before dedup"), (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[{name}]")
    want_sam = subprocess.run([common.ORACLE, "mem", "-t", "8", *flags, small_index["prefix"], fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got_sam = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    assert _body(got_sam) == _body(want_sam), f"SAM differs under {flags}"
    if name in ("W1", "W5", "W3_k17_w50"):                     # 22*W <= read length: the seed SW filter ran and re-scored seeds
        base = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_CHAIN_FLT]))
        assert any(not np.array_equal(g[bw.STAGE_CHAIN_FLT], b[bw.STAGE_CHAIN_FLT]) for g, b in zip(got, base)), "-W changed nothing"


@pytest.mark.parametrize("name", sorted(common.PE_OPTION_SETS))
def test_pe_sam_under_options(ctx, small_index, se_reads, name):
    flags = common.PE_OPTION_SETS[name]
    opt, pes0 = common.opt_from_cli(flags)
    opt.n_threads = 8
    opt.flag |= 0x2
    f1, f2 = se_reads["pe"]
    n1, s1, q1 = bw.read_fastq(f1)
    n2, s2, q2 = bw.read_fastq(f2)
    names = [x for p in zip(n1, n2) for x in p]
    seqs = [x for p in zip(s1, s2) for x in p]
    quals = [x for p in zip(q1, q2) for x in p]
    want = subprocess.run([common.ORACLE, "mem", "-t", "8", *flags, small_index["prefix"], f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got = b"".join(ctx.process_seqs(names, seqs, quals, opt, pes0=pes0))
    assert _body(got) == _body(want), f"PE SAM differs under {flags}"


@pytest.mark.parametrize("name", sorted(common.LONG_OPTION_SETS))
def test_long_reads_seed_sw_filter(ctx, small_index, tmp_path, name):
    """600-700 base reads: with -W 20..31 mem_flt_chained_seeds is active AND its threshold exceeds the score of bare
    short seeds, so seeds are dropped; W = 0 is inactive below ~730 bases.  Stages and SAM vs the CPU path."""
    flags = common.LONG_OPTION_SETS[name]
    opt, _ = common.opt_from_cli(flags)
    opt.n_threads = 8
    fq = str(tmp_path / "long.fq")
    parts = []
    for k, ln in enumerate((600, 650, 700)):
        p = str(tmp_path / f"l{k}.fq")
        bw.make_reads(small_index["fa"], p, None, 120, ln, 40000, 3000, 500, 720 + k, 50000)
        parts.append(open(p).read().replace("@r", f"@l{k}_"))
    open(fq, "w").write("".join(parts))
    names, seqs, quals = bw.read_fastq(fq)
    obin = str(tmp_path / "o.bin")
    subprocess.check_call([common.ORACLE, "stages", *flags, small_index["prefix"], fq, obin])
    want = common.by_read(bw.read_record_file(obin))
    codes, off = bw.pack_reads(seqs)
    stages = [bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS]
    got = common.by_read(ctx.run_stages(codes, off, stages, opt))
    for st, what in [(bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS_PRE, "regions before dedup"), (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[{name}]")
    want_sam = subprocess.run([common.ORACLE, "mem", "-t", "8", *flags, small_index["prefix"], fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert _body(b"".join(ctx.process_seqs(names, seqs, quals, opt))) == _body(want_sam), f"SAM differs under {flags}"
    if flags and flags[0] == "-W":
        # seeds were really dropped: fewer seeds in the kept chains than without the SW filter's threshold (-W 1 keeps all)
        o1, _ = common.opt_from_cli(["-W", "1"] + flags[2:])
        base = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_CHAIN_FLT], o1))
        n_seeds = lambda recs: sum(int(sum(_chain_seed_counts(r[bw.STAGE_CHAIN_FLT]))) for r in recs)
        assert n_seeds(got) < n_seeds(base), "no seed was dropped by the SW filter"


def _chain_seed_counts(rec):
    n, i, out = int(rec[0]), 1, []
    for _ in range(n):
        out.append(int(rec[i + 7]))
        i += 8 + 4 * int(rec[i + 7])
    return out


def test_golden_option_sweep_from_reference(built, tmp_path):
    """HIP path vs SAM the REFERENCE itself produced under non-default options (tests/golden/opt_*.sam.gz)."""
    import gzip
    import os
    G = common.GOLDEN
    fa = str(tmp_path / "g60k.fa")
    open(fa, "wb").write(gzip.open(os.path.join(G, "g60k.fa.gz")).read())
    bw.make_index(fa, str(tmp_path / "g60k"))
    open(str(tmp_path / "g60k.alt"), "wb").write(open(os.path.join(G, "g60k.alt"), "rb").read())
    for n in ("se.fq", "pe_1.fq", "pe_2.fq", "long.fq"):
        open(str(tmp_path / n), "wb").write(gzip.open(os.path.join(G, n + ".gz")).read())
    se = bw.read_fastq(str(tmp_path / "se.fq"))
    lg = bw.read_fastq(str(tmp_path / "long.fq"))
    n1, s1, q1 = bw.read_fastq(str(tmp_path / "pe_1.fq"))
    n2, s2, q2 = bw.read_fastq(str(tmp_path / "pe_2.fq"))
    pe = ([x for p in zip(n1, n2) for x in p], [x for p in zip(s1, s2) for x in p], [x for p in zip(q1, q2) for x in p])
    with bw.Context(str(tmp_path / "g60k")) as c:
        for name in common.GOLDEN_OPTION_SETS:
            opt, pes0 = common.opt_from_cli(common.option_flags(name))
            opt.n_threads = 4
            reads = pe if name in common.PE_OPTION_SETS else lg if name in common.LONG_OPTION_SETS else se
            if name in common.PE_OPTION_SETS:
                opt.flag |= 0x2
            got = b"".join(c.process_seqs(*reads, opt, pes0=pes0))
            want = gzip.open(os.path.join(G, f"opt_{name}.sam.gz")).read()
            assert got == want, f"{name}: SAM differs from the reference's"


def test_ksw_align2_known_answers_from_reference(ctx):
    """Device ksw_align2 (lane-exact striped SW, byte and word kernels, start recovery by the reversed second pass,
    second-best score) vs the reference's own results in tests/golden/kat_ksw.npz (as mem_matesw calls it) and
    tests/golden/kat_ksw_align.npz (word kernel as mem_seed_sw calls it, other matrices and gap costs)."""
    import os
    params, qs, ts, want, mats = [], [], [], [], []
    for tag, v in bw.parse_records(np.load(os.path.join(common.GOLDEN, "kat_ksw.npz"))["words"]):
        if tag != 22:
            continue
        v = [int(x) for x in v]
        qlen, tlen = v[0], v[1]
        xtra = 0x40000 | 0x80000 | (0x10000 if qlen < 250 else 0) | 19
        params.append([qlen, tlen, xtra, v[6], v[7], v[8], v[9], 0]); qs.append(v[10:10 + qlen]); ts.append(v[10 + qlen:10 + qlen + tlen])
        want.append(v[10 + qlen + tlen:])
    qoff = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int64)
    toff = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.int64)
    got = ctx.kat_ksw_align(np.array(params), np.concatenate(qs), qoff, np.concatenate(ts), toff)
    assert len(want) >= 100
    bad = [i for i in range(len(want)) if list(got[i]) != want[i]]
    assert not bad, f"{len(bad)} of {len(want)} differ; first {bad[0]}: params={params[bad[0]]} got={list(got[bad[0]])} want={want[bad[0]]}"
    path = os.path.join(common.GOLDEN, "kat_ksw_align.npz")
    groups = {}
    for tag, v in bw.parse_records(np.load(path)["words"]):
        if tag != 23:
            continue
        v = [int(x) for x in v]
        qlen, tlen, xtra = v[0], v[1], v[2]
        mat = tuple(v[7:32])
        g = groups.setdefault(mat, ([], [], [], []))
        g[0].append([qlen, tlen, xtra, v[3], v[4], v[5], v[6], 0]); g[1].append(v[32:32 + qlen]); g[2].append(v[32 + qlen:32 + qlen + tlen])
        g[3].append(v[32 + qlen + tlen:])
    total = 0
    for mat, (params, qs, ts, want) in groups.items():
        qoff = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int64)
        toff = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.int64)
        got = ctx.kat_ksw_align(np.array(params), np.concatenate(qs), qoff, np.concatenate(ts), toff, mat=np.array(mat, dtype=np.int8))
        bad = [i for i in range(len(want)) if list(got[i]) != want[i]]
        assert not bad, f"mat {mat[:6]}..: {len(bad)} of {len(want)} differ; first {bad[0]}: params={params[bad[0]]} got={list(got[bad[0]])} want={want[bad[0]]}"
        total += len(want)
    assert total >= 300
