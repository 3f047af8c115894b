"""GPU parity tests proper: every stage boundary of mem_align1_core, HIP path (through the C ABI)
vs the oracle on the same seeded reads.  Bit-exact (integer/index work)."""
import os
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


KNOB_DEFAULTS = dict(intv_cap=96, smem_lanes=1, heavy_mult=10, chain_big_min=512, rank_sort_min=2, spec_min_chains=16, ext_lds_window=1 << 30)


@pytest.fixture
def tuned(ctx):
    """ctx.tune(...) for one test; the defaults come back afterwards (the context is shared by the module)."""
    yield ctx.tune
    ctx.tune(**KNOB_DEFAULTS)


def _reads(small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim=0):
    fq = str(tmp_path / f"{name}.fq")
    bw.make_reads(small_index["fa"], fq, None, n, length, sub, indel, nn, seed, chim)
    _, seqs, _ = bw.read_fastq(fq)
    return fq, seqs


@pytest.mark.parametrize("name,n,length,sub,indel,nn,seed,chim", [
    ("se150", 3000, 150, 10000, 2000, 500, 101, 20000),
    ("se100", 2000, 100, 10000, 0, 0, 102, 0),
    ("se250", 1500, 250, 50000, 3000, 500, 105, 30000),
    ("short", 500, 30, 20000, 0, 20000, 106, 0),
])
def test_intervals_match_oracle(ctx, small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim):
    fq, seqs = _reads(small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV]))
    common.assert_stage_equal(got, want, bw.STAGE_INTV, f"intervals[{name}]")


@pytest.mark.parametrize("lanes,heavy_mult", [(1, 10), (2, 10), (4, 10), (8, 10), (2, 0), (2, 1), (1, 2), (4, 3)])
def test_intervals_every_smem_variant(ctx, small_index, tmp_path, tuned, lanes, heavy_mult):
    """Every lanes-per-read variant of k_smem, and k_smem_heavy forced onto most reads (heavy_mult 1..3: a read is
    handed over after heavy_mult x len bwt_extend calls; 0 = never), must give the oracle's interval lists."""
    tuned(smem_lanes=lanes, heavy_mult=heavy_mult)
    fq, seqs = _reads(small_index, tmp_path, "variants", 2500, 150, 10000, 2000, 2000, 111, 20000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV]))
    common.assert_stage_equal(got, want, bw.STAGE_INTV, f"intervals[lanes={lanes},heavy_mult={heavy_mult}]")
    if heavy_mult in (1, 2, 3):
        assert ctx.counters()["heavy_intv"] > 0, "k_smem_heavy did not run"


MASK_CHAIN_PRE = True


def _mask_chain_prefilter(rec):
    """w / kept / first are uninitialised in the reference before mem_chain_flt: ignore them."""
    a = rec.copy()
    n, i = int(a[0]), 1
    for _ in range(n):
        a[i + 3] = a[i + 4] = a[i + 5] = 0
        i += 8 + 4 * int(a[i + 7])
    return a


@pytest.mark.parametrize("name,n,length,sub,indel,nn,seed,chim", [
    ("se150", 3000, 150, 10000, 2000, 500, 101, 20000),
    ("se100", 2000, 100, 10000, 0, 0, 102, 0),
    ("se250", 1500, 250, 50000, 3000, 500, 105, 30000),
    ("short", 500, 30, 20000, 0, 20000, 106, 0),
    ("se600", 300, 600, 30000, 3000, 500, 107, 30000),
])
def test_all_stages_match_oracle(ctx, small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim):
    fq, seqs = _reads(small_index, tmp_path, name, n, length, sub, indel, nn, seed, chim)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    stages = [bw.STAGE_INTV, bw.STAGE_CHAIN, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS]
    got = common.by_read(ctx.run_stages(codes, off, stages))
    for g, w in zip(got, want):
        g[bw.STAGE_CHAIN] = _mask_chain_prefilter(g[bw.STAGE_CHAIN])
        w[bw.STAGE_CHAIN] = _mask_chain_prefilter(w[bw.STAGE_CHAIN])
    for st, what in [(bw.STAGE_INTV, "intervals"), (bw.STAGE_CHAIN, "chains"), (bw.STAGE_CHAIN_FLT, "filtered chains"),
                     (bw.STAGE_REGS_PRE, "regions before dedup"), (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[{name}]")


@pytest.mark.parametrize("rank_min", [2, 1 << 30])
def test_dedup_with_forced_rank_sort(ctx, small_index, tmp_path, tuned, rank_min):
    """mem_sort_dedup_patch: the wavefront rank sort (taken when no two keys are equal, else the exact one-lane introsort)
    forced onto every list of >= 2 regions, and switched off: regions after dedup must not change."""
    tuned(rank_sort_min=rank_min)
    fq, seqs = _reads(small_index, tmp_path, "rank", 3000, 150, 20000, 3000, 500, 123, 30000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_REGS_PRE, bw.STAGE_REGS]))
    common.assert_stage_equal(got, want, bw.STAGE_REGS, f"regions[rank_sort_min={rank_min}]")


@pytest.mark.parametrize("spec_min", [0, 1, 2])
def test_regions_with_forced_ahead_of_time_extension(ctx, small_index, tmp_path, tuned, spec_min):
    """k_extend_spec (best seed of each chain extended by its own wavefront before k_extend decides) forced onto every
    read with >= spec_min chains (1: all reads; 0: switched off): regions before and after dedup must not change."""
    tuned(spec_min_chains=spec_min)
    fq, seqs = _reads(small_index, tmp_path, "spec", 3000, 150, 20000, 3000, 500, 113, 30000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_REGS_PRE, bw.STAGE_REGS]))
    common.assert_stage_equal(got, want, bw.STAGE_REGS_PRE, f"regions before dedup[spec_min={spec_min}]")
    common.assert_stage_equal(got, want, bw.STAGE_REGS, f"regions[spec_min={spec_min}]")


@pytest.mark.parametrize("window", [150, 400])
def test_regions_with_forced_large_window_variant(ctx, small_index, tmp_path, tuned, window):
    """k_extend_big (reference window of a chain in a global-memory slab instead of LDS; taken by reads whose chains drift
    far, e.g. in tandem repeats, or under a wide -w) forced onto ordinary reads by shrinking the LDS window: every read
    with a chain window above `window` bases is handed over; regions before and after dedup must not change."""
    tuned(ext_lds_window=window)
    fq, seqs = _reads(small_index, tmp_path, "bigwin", 2500, 150, 20000, 3000, 500, 127, 30000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_REGS_PRE, bw.STAGE_REGS]))
    common.assert_stage_equal(got, want, bw.STAGE_REGS_PRE, f"regions before dedup[lds_window={window}]")
    common.assert_stage_equal(got, want, bw.STAGE_REGS, f"regions[lds_window={window}]")


def test_tandem_repeat_reads_with_drifting_chains(built, tmp_path):
    """Reads from a long, slightly diverged tandem array: chains collect many seeds whose diagonals drift, so the chain's
    reference window grows far beyond read length + 2 gaps (the LDS window of k_extend); such reads must go through
    k_extend_big and still equal the CPU path, also under -w 300."""
    import subprocess
    rng = np.random.default_rng(17)
    unit = rng.integers(0, 4, 37)
    arr = []
    for k in range(400):                                             # 400 copies of a 37-mer, 3 % substitutions, occasional 1-base indels
        u = unit.copy()
        m = rng.random(37) < 0.03
        u[m] = (u[m] + rng.integers(1, 4, m.sum())) % 4
        u = list(u)
        if rng.random() < 0.15:
            del u[int(rng.integers(0, len(u)))]
        if rng.random() < 0.15:
            u.insert(int(rng.integers(0, len(u))), int(rng.integers(0, 4)))
        arr += u
    flank = lambda n: list(rng.integers(0, 4, n))
    g = flank(20000) + arr + flank(20000)
    seq = "".join("ACGT"[x] for x in g)
    fa = str(tmp_path / "tr.fa")
    with open(fa, "w") as f:
        f.write(">tr\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + "\n")
    prefix = str(tmp_path / "tr")
    bw.make_index(fa, prefix)
    reads = []
    for i in range(600):                                             # reads inside and across the edges of the array
        st = int(rng.integers(19000, 20000 + len(arr) + 800))
        ln = int(rng.choice([150, 250, 600]))
        r = list(g[st:st + ln])
        for j in range(len(r)):
            if rng.random() < 0.02:
                r[j] = (r[j] + int(rng.integers(1, 4))) % 4
        s = "".join("ACGT"[x] for x in r)
        if i & 1:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        reads.append(s)
    fq = str(tmp_path / "tr.fq")
    with open(fq, "w") as f:
        for i, s in enumerate(reads):
            f.write(f"@t{i}\n{s}\n+\n{'I' * len(s)}\n")
    names, seqs, quals = bw.read_fastq(fq)
    codes, off = bw.pack_reads(seqs)
    with bw.Context(prefix) as c:
        for flags in ([], ["-w", "300"]):
            opt, _ = common.opt_from_cli(flags)
            opt.n_threads = 4
            obin = str(tmp_path / "o.bin")
            subprocess.check_call([common.ORACLE, "stages", *flags, prefix, fq, obin])
            want = common.by_read(bw.read_record_file(obin))
            got = common.by_read(c.run_stages(codes, off, [bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS], opt))
            for st, what in [(bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS_PRE, "regions before dedup"), (bw.STAGE_REGS, "regions")]:
                common.assert_stage_equal(got, want, st, f"{what}[tandem {flags}]")
            want_sam = subprocess.run([common.ORACLE, "mem", "-t", "4", *flags, prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            assert b"".join(c.process_seqs(names, seqs, quals, opt)) == want_sam, f"tandem SAM {flags}"


_ADOPT_SCRIPT = r"""
import sys, os
import numpy as np
import torch                                  # first, as in bench.py: torch brings its own HIP runtime into the process
torch.cuda.set_device(0)
root, prefix, fq, obin = sys.argv[1:5]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "bwa-mem-gpu_amd"))
import common
from common import bw
import tools_py as tp
meta, arrays = tp.load_index_arrays(prefix)
tensors = {k: torch.from_numpy(v).to("cuda:0") for k, v in arrays.items()}
torch.cuda.synchronize()
ctx = bw.Context.from_device_arrays(meta, tensors["bwt"].data_ptr(), tensors["sa"].data_ptr(), tensors["pac"].data_ptr(), 0)
_, seqs, _ = bw.read_fastq(fq)
want = common.by_read(common.oracle_stages(prefix, fq, obin))
codes, off = bw.pack_reads(seqs)
got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV, bw.STAGE_REGS]))
common.assert_stage_equal(got, want, bw.STAGE_INTV, "intervals[adopted index]")
common.assert_stage_equal(got, want, bw.STAGE_REGS, "regions[adopted index]")
ctx.close()
print("ADOPT_OK", len(got))
"""


def test_context_on_adopted_device_arrays(small_index, tmp_path):
    """bwahip_init_device: the index arrays already sit in HBM as torch tensors (what every rank but 0 holds after the
    RCCL broadcast in bench.py); the context adopts them zero-copy and must give the oracle's results.  Own process,
    torch initialised first, exactly like a bench.py rank."""
    import subprocess, sys
    fq, _ = _reads(small_index, tmp_path, "adopt", 1500, 150, 10000, 2000, 500, 121, 20000)
    script = tmp_path / "adopt.py"
    script.write_text(_ADOPT_SCRIPT)
    r = subprocess.run([sys.executable, str(script), common.ROOT, small_index["prefix"], fq, str(tmp_path / "o.bin")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ADOPT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_interval_list_overflow_is_rerun_on_the_gpu(ctx, small_index, tmp_path, tuned):
    """A per-read interval capacity that is too small must be detected by k_smem and the batch re-run with more room
    (no CPU path): start with room for 3 intervals per read."""
    tuned(intv_cap=3)
    fq, seqs = _reads(small_index, tmp_path, "ovf", 1500, 150, 20000, 3000, 500, 119, 30000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV, bw.STAGE_REGS]))
    common.assert_stage_equal(got, want, bw.STAGE_INTV, "intervals[cap=3]")
    common.assert_stage_equal(got, want, bw.STAGE_REGS, "regions[cap=3]")
    assert max(len(g[bw.STAGE_INTV]) for g in got) > 3 * 4


@pytest.mark.parametrize("big_min", [-1, 0, 8])
def test_chains_with_forced_lds_btree(ctx, small_index, tmp_path, tuned, big_min):
    """k_chain_big (B-tree nodes in LDS, one read per workgroup) forced onto every read with more than big_min seeds
    (0: all reads with seeds; -1: switched off): chains before and after filtering must not change."""
    tuned(chain_big_min=big_min)
    fq, seqs = _reads(small_index, tmp_path, "big", 3000, 150, 20000, 3000, 500, 117, 30000)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_CHAIN, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS]))
    for g, w in zip(got, want):
        g[bw.STAGE_CHAIN] = _mask_chain_prefilter(g[bw.STAGE_CHAIN])
        w[bw.STAGE_CHAIN] = _mask_chain_prefilter(w[bw.STAGE_CHAIN])
    for st, what in [(bw.STAGE_CHAIN, "chains"), (bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[big_min={big_min}]")


def test_fm_known_answers_vs_oracle_lib(ctx, small_index):
    """Device Occ / SA / extend against the oracle's C functions on random rows."""
    import ctypes as C
    ora = C.CDLL(os.path.join(common.ROOT, "oracle", "liboracle.so"))
    ora.ora_index_load.restype = C.c_void_p
    ora.ora_index_load.argtypes = [C.c_char_p]
    idx = ora.ora_index_load(small_index["prefix"].encode())
    fmi = C.cast(idx, C.POINTER(C.c_void_p))[0]
    ora.ora_occ4.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    ora.ora_sa.argtypes = [C.c_void_p, C.c_uint64]
    ora.ora_sa.restype = C.c_uint64
    seq_len = 2 * 1000000
    rng = np.random.default_rng(5)
    k = np.concatenate([np.array([2**64 - 1, seq_len, 0, 1, 127, 128, 129], dtype=np.uint64),
                        rng.integers(0, seq_len + 1, 4000, dtype=np.uint64)])
    got = ctx.kat_occ4(k)
    want = np.zeros_like(got)
    buf = (C.c_uint64 * 4)()
    for i, kk in enumerate(k):
        ora.ora_occ4(fmi, int(kk), buf)
        want[i] = list(buf)
    assert np.array_equal(got, want)
    ks = k[1:]
    got_sa = ctx.kat_sa(ks)
    want_sa = np.array([ora.ora_sa(fmi, int(kk)) for kk in ks], dtype=np.uint64)
    assert np.array_equal(got_sa, want_sa)


def test_sa_table_densities_agree(small_index, tmp_path, monkeypatch):
    """The SA table filled in on the GPU (every row, every 4th) returns for EVERY BWT row what the walk over the index files' every-32nd
    table returns (bwt_sa, bwt.c:86); a clone shares the table; the SAM text does not depend on the density."""
    seq_len = 2 * 1000000
    rows = np.arange(0, seq_len + 1, dtype=np.uint64)
    fq, seqs = _reads(small_index, tmp_path, "sa_dens", 600, 150, 10000, 2000, 500, 77)
    names, _, quals = bw.read_fastq(fq)
    got, sams = {}, {}
    for intv in (32, 4, 1):
        monkeypatch.setenv("BWAHIP_SA_INTV", str(intv))
        with bw.Context(small_index["prefix"], 0) as c:
            fp = c.index_footprint()
            assert fp["sa_intv"] == intv and fp["sa_gb"] == round((seq_len // intv + 1) * 8 / 1e9, 2)
            got[intv] = c.kat_sa(rows)
            sams[intv] = c.process_seqs(names, seqs, quals)
            if intv == 1:
                with c.clone() as c2:
                    assert np.array_equal(c2.kat_sa(rows), got[intv])
    assert np.array_equal(np.sort(got[32][1:]), np.arange(seq_len, dtype=np.uint64))     # a permutation of the text positions
    assert got[32][0] == np.uint64(2**64 - 1) or got[32][0] == seq_len                   # row 0: the sentinel's suffix as bwt_sa returns it
    assert np.array_equal(got[4], got[32]) and np.array_equal(got[1], got[32])
    assert sams[4] == sams[32] and sams[1] == sams[32]


def test_interval_table_entries_equal_forward_extension(ctx, small_index, tmp_path, monkeypatch):
    """The interval table of the BWT search is filled by BACKWARD extensions; every entry of every length must equal the FORWARD extension
    (bwt_extend, bwt.c:262, is_back = 0) of the entry of the string without its last base, i.e. the bi-interval is a function of the string
    alone -- and the SAM text with pass 3 (bwt_seed_strategy1) jumping through the table (any K) is the text without it."""
    k, bad = ctx.kat_kmer_table()
    assert k >= 8 and bad == 0
    fq, seqs = _reads(small_index, tmp_path, "kmer_tab", 800, 150, 10000, 2000, 2000, 78)
    names, _, quals = bw.read_fastq(fq)
    sams = {}
    for K in (0, 3, 7, 16):
        monkeypatch.setenv("BWAHIP_KMER_K", str(K))
        with bw.Context(small_index["prefix"], 0) as c:
            kk, bad = c.kat_kmer_table()
            assert bad == 0 and kk == (0 if K == 0 else min(K, 11))
            sams[K] = c.process_seqs(names, seqs, quals)
            cnt = c.counters()
            assert (cnt["pass3_jumped"] > 0) == (K >= 2)
    assert sams[3] == sams[0] and sams[7] == sams[0] and sams[16] == sams[0]
    assert ctx.process_seqs(names, seqs, quals) == sams[0]


def test_extend_known_answers_vs_oracle_lib(ctx, small_index):
    """bwt_extend on the device for random walks (both directions), incl. size-1 intervals."""
    import ctypes as C

    class Intv(C.Structure):
        _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]
    ora = C.CDLL(os.path.join(common.ROOT, "oracle", "liboracle.so"))
    ora.ora_index_load.restype = C.c_void_p
    ora.ora_index_load.argtypes = [C.c_char_p]
    idx = ora.ora_index_load(small_index["prefix"].encode())
    fmi = C.cast(idx, C.POINTER(C.c_void_p))[0]
    ora.ora_extend.argtypes = [C.c_void_p, C.POINTER(Intv), C.POINTER(Intv), C.c_int]
    ora.ora_set_intv.argtypes = [C.c_void_p, C.c_int, C.POINTER(Intv)]
    rng = np.random.default_rng(7)
    iks, backs, wants = [], [], []
    ok = (Intv * 4)()
    for walk in range(300):
        ik = Intv()
        ora.ora_set_intv(fmi, int(rng.integers(0, 4)), C.byref(ik))
        for step in range(40):
            if ik.x[2] == 0:
                break
            back = int(rng.integers(0, 2))
            ora.ora_extend(fmi, C.byref(ik), ok, back)
            iks.append([ik.x[0], ik.x[1], ik.x[2]])
            backs.append(back)
            wants.append([v for o in ok for v in (o.x[0], o.x[1], o.x[2])])
            c = int(rng.integers(0, 4))
            best = max(range(4), key=lambda b: ok[b].x[2]) if step % 3 else c   # mostly follow a surviving path
            nxt = Intv()
            nxt.x[0], nxt.x[1], nxt.x[2] = ok[best].x[0], ok[best].x[1], ok[best].x[2]
            ik = nxt
    got = ctx.kat_extend(np.array(iks, dtype=np.uint64), np.array(backs, dtype=np.int32))
    want = np.array(wants, dtype=np.uint64)
    sizes = np.array(iks, dtype=np.uint64)[:, 2]
    assert (sizes == 1).sum() > 100
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {len(want)} extends differ; first: ik={iks[bad[0]]} back={backs[bad[0]]} got={got[bad[0]]} want={want[bad[0]]}"


def test_ksw_extend_known_answers_from_reference(ctx):
    """Wavefront ksw_extend2 vs the reference's own results (tests/golden/kat_ksw.npz): narrow and wide bands,
    z-drop on/off, asymmetric gap costs, ambiguous bases, query lengths 1..250."""
    words = np.load(os.path.join(common.GOLDEN, "kat_ksw.npz"))["words"]
    params, qs, ts, want = [], [], [], []
    for tag, v in bw.parse_records(words):
        if tag != 20:
            continue
        v = [int(x) for x in v]
        qlen, tlen = v[0], v[1]
        params.append(v[:10]); qs.append(v[10:10 + qlen]); ts.append(v[10 + qlen:10 + qlen + tlen]); want.append(v[10 + qlen + tlen:])
    qoff = np.concatenate([[0], np.cumsum([len(x) for x in qs])]).astype(np.int64)
    toff = np.concatenate([[0], np.cumsum([len(x) for x in ts])]).astype(np.int64)
    got = ctx.kat_ksw_extend(np.array(params), np.concatenate(qs), qoff, np.concatenate(ts), toff)
    assert len(want) >= 100
    bad = [i for i in range(len(want)) if list(got[i]) != want[i]]
    assert not bad, f"{len(bad)} of {len(want)} differ; first {bad[0]}: params={params[bad[0]]} got={list(got[bad[0]])} want={want[bad[0]]}"


def _genome_slice(small_index, start, length):
    seq = []
    with open(small_index["fa"], "rb") as f:
        f.readline()
        for line in f:
            if line.startswith(b">"):
                break
            seq.append(line.strip())
    g = b"".join(seq)
    return g[start:start + length]


def test_wavefront_introsort_equals_ksort_restatement(ctx):
    """The region-list sorts are the reference's UNSTABLE ks_introsort (ksort.h:176): with equal keys the resulting order is a property of that
    algorithm.  The kernels run it on the whole wavefront (csrc/isort_dev.h: parallel partition steps + stable final placement); here its
    permutation must equal the one-lane restatement's on keys with no, few and many ties, in both key modes, and must be a sorted order."""
    rng = np.random.default_rng(77)
    n_par = 0
    for trial in range(160):
        n = int(rng.choice([1, 2, 3, 5, 16, 17, 18, 33, 63, 64, 65, 100, 129, 257, 700, 1500, 4000]))
        mode = trial & 1
        kind = trial % 5
        if kind == 0:                                            # no ties
            k64 = rng.permutation(n).astype(np.int64) * 7919
            score = rng.integers(20, 150, n).astype(np.int32); qb = rng.integers(0, 100, n).astype(np.int32)
        elif kind == 1:                                          # a few duplicates (a rescued hit that repeats one of the list)
            k64 = rng.permutation(n).astype(np.int64) * 13
            score = rng.integers(20, 150, n).astype(np.int32); qb = rng.integers(0, 100, n).astype(np.int32)
            for _ in range(max(1, n // 20)):
                i, j = rng.integers(0, n, 2)
                k64[i], score[i], qb[i] = k64[j], score[j], qb[j]
        elif kind == 2:                                          # many ties
            k64 = rng.integers(0, max(1, n // 4), n).astype(np.int64)
            score = rng.integers(0, 3, n).astype(np.int32); qb = rng.integers(0, 2, n).astype(np.int32)
        elif kind == 3:                                          # three distinct keys
            k64 = rng.integers(0, 3, n).astype(np.int64); score = np.zeros(n, np.int32); qb = np.zeros(n, np.int32)
        else:                                                    # sorted input with ties (the reference's introsort reaches its depth limit here)
            k64 = np.sort(rng.integers(0, max(1, n // 2), n)).astype(np.int64); score = np.full(n, 60, np.int32); qb = np.zeros(n, np.int32)
        par, seq, ran = ctx.kat_introsort(k64, score, qb, mode)
        assert sorted(seq.tolist()) == list(range(n))
        key = (lambda i: (int(k64[i]),)) if mode == 0 else (lambda i: (-int(score[i]), int(k64[i]), int(qb[i])))
        assert all(key(seq[i]) <= key(seq[i + 1]) for i in range(n - 1)), "the restatement did not sort"
        if ran:
            n_par += 1
            assert np.array_equal(par, seq), f"trial {trial}: n {n} mode {mode} kind {kind}"
        else:
            assert kind == 4 or n > 1000, f"unexpected hand-over: n {n} kind {kind}"   # only the depth limit hands over
    assert n_par >= 100


def test_edge_cases_ragged_batch(ctx, small_index, tmp_path):
    """Empty batch; one batch mixing an empty read, reads shorter than a seed, an all-N read, a read with N runs, a read
    of the maximum length (700) and ordinary ones: stages and SAM must equal the CPU path; 701 bases is a loud error."""
    import subprocess
    assert ctx.run_stages(np.zeros(0, np.uint8), np.zeros(1, np.int64), [bw.STAGE_INTV, bw.STAGE_REGS]) == []      # n = 0
    g = _genome_slice(small_index, 1000, 4000)
    reads = [b"", b"A", g[10:28], g[100:119], b"N" * 150, g[200:260] + b"NNNNN" + g[265:350], g[500:1200], g[1300:1450],
             g[2000:2150].translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], b"ACGT" * 40]
    assert len(reads[6]) == 700
    fq = str(tmp_path / "edge.fq")
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@e%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    names, seqs, quals = bw.read_fastq(fq)
    want = common.by_read(common.oracle_stages(small_index["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS]))
    for st, what in [(bw.STAGE_INTV, "intervals"), (bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[edge cases]")
    want_sam = subprocess.run([common.ORACLE, "mem", "-t", "2", small_index["prefix"], fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    opt = bw.default_opt()
    opt.n_threads = 2
    got_sam = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
    assert body(got_sam) == body(want_sam)
    too_long = [g[0:701]]
    codes, off = bw.pack_reads(too_long)
    with pytest.raises(bw.BwahipError, match="ECAPACITY"):
        ctx.run_stages(codes, off, [bw.STAGE_REGS])
