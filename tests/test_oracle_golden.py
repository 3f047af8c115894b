"""CPU tests: the oracle (plain-C restatement) against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py ran oracle/_ref/bwaref, i.e. the reference's own sources).  Bit-exact."""
import ctypes as C
import gzip
import hashlib
import os
import subprocess
import numpy as np
import pytest
import common
from common import bw

G = common.GOLDEN


@pytest.fixture(scope="module")
def gold(built, tmp_path_factory):
    d = tmp_path_factory.mktemp("gold")
    fa = str(d / "g60k.fa")
    open(fa, "wb").write(gzip.open(os.path.join(G, "g60k.fa.gz")).read())
    bw.make_index(fa, str(d / "g60k"))            # product tool; checked against the reference's hashes below
    open(str(d / "g60k.alt"), "wb").write(open(os.path.join(G, "g60k.alt"), "rb").read())
    for n in ("se.fq", "pe_1.fq", "pe_2.fq"):
        open(str(d / n), "wb").write(gzip.open(os.path.join(G, n + ".gz")).read())
    return {"dir": str(d), "prefix": str(d / "g60k"), "fa": fa}


def test_mkindex_matches_reference_index_files(gold):
    """mkindex must write byte-identical .pac/.ann/.amb/.bwt/.sa to the reference's `bwa index`."""
    want = dict(reversed(l.split()) for l in open(os.path.join(G, "index.sha256")))
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        got = hashlib.sha256(open(f"{gold['prefix']}.{ext}", "rb").read()).hexdigest()
        assert got == want[f"g60k.{ext}"], f".{ext} differs from the reference's index"


def _run_oracle(args):
    return subprocess.run([common.ORACLE] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout


@pytest.mark.parametrize("name,extra,fqs", [("se", [], ["se.fq"]), ("se_all", ["-a"], ["se.fq"]), ("pe", [], ["pe_1.fq", "pe_2.fq"])])
def test_oracle_sam_equals_reference_sam(gold, name, extra, fqs):
    want = gzip.open(os.path.join(G, name + ".sam.gz")).read()
    got = _run_oracle(["mem", "-t", "4"] + extra + [gold["prefix"]] + [os.path.join(gold["dir"], f) for f in fqs])
    assert got == want


def test_oracle_sam_independent_of_threads_and_batching(gold):
    want = gzip.open(os.path.join(G, "se.sam.gz")).read()
    fq = os.path.join(gold["dir"], "se.fq")
    assert _run_oracle(["mem", "-t", "1", gold["prefix"], fq]) == want
    assert _run_oracle(["mem", "-t", "3", "-K", "5000", gold["prefix"], fq]) == want


def test_oracle_stage_dump_equals_reference(gold, tmp_path):
    want = np.load(os.path.join(G, "se.stages.npz"))["words"]
    out = str(tmp_path / "o.bin")
    subprocess.check_call([common.ORACLE, "stages", gold["prefix"], os.path.join(gold["dir"], "se.fq"), out])
    got = np.fromfile(out, dtype=np.int64)
    assert np.array_equal(got, want)


class Intv(C.Structure):
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


@pytest.fixture(scope="module")
def ora(gold):
    lib = C.CDLL(os.path.join(common.ROOT, "oracle", "liboracle.so"))
    lib.ora_index_load.restype = C.c_void_p
    lib.ora_index_load.argtypes = [C.c_char_p]
    idx = lib.ora_index_load(gold["prefix"].encode())
    fmi = C.cast(idx, C.POINTER(C.c_void_p))[0]
    lib.ora_occ4.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ora_sa.argtypes = [C.c_void_p, C.c_uint64]
    lib.ora_sa.restype = C.c_uint64
    lib.ora_extend.argtypes = [C.c_void_p, C.POINTER(Intv), C.POINTER(Intv), C.c_int]
    return lib, fmi


def test_fm_known_answers(ora):
    """Occ / SA / bwt_extend known answers from the reference (incl. k=-1, k=seq_len, k=primary)."""
    lib, fmi = ora
    n_occ = n_sa = n_ext = 0
    buf = (C.c_uint64 * 4)()
    ok = (Intv * 4)()
    for tag, v in bw.parse_records(np.load(os.path.join(G, "kat_fm.npz"))["words"]):
        v = [int(x) & (2**64 - 1) for x in v]
        if tag == 10:
            lib.ora_occ4(fmi, v[0], buf)
            assert list(buf) == v[1:5]
            n_occ += 1
        elif tag == 12:
            assert lib.ora_sa(fmi, v[0]) == v[1]
            n_sa += 1
        elif tag == 11:
            ik = Intv()
            ik.x[0], ik.x[1], ik.x[2] = v[0], v[1], v[2]
            lib.ora_extend(fmi, C.byref(ik), ok, v[3])
            assert [x for o in ok for x in (o.x[0], o.x[1], o.x[2])] == v[4:16]
            n_ext += 1
    assert n_occ >= 150 and n_sa >= 149 and n_ext > 1000


def test_ksw_known_answers():
    """ksw_extend2 / ksw_global2 (score + CIGAR) / ksw_align2 known answers from the reference: band edges,
    z-drop on and off, asymmetric gap costs, ambiguous bases."""
    lib = C.CDLL(os.path.join(common.ROOT, "oracle", "liboracle.so"))
    mat = (C.c_int8 * 25)()
    lib.ora_fill_scmat(1, 4, mat)
    u8p, ip = C.POINTER(C.c_uint8), C.POINTER(C.c_int)
    lib.ora_ksw_extend2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int8)] + [C.c_int] * 8 + [ip] * 5
    lib.ora_ksw_global2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int8)] + [C.c_int] * 5 + [ip, C.POINTER(C.POINTER(C.c_uint32))]

    class Kswr(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]
    lib.ora_ksw_align2.restype = Kswr
    lib.ora_ksw_align2.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int8)] + [C.c_int] * 5
    counts = {20: 0, 21: 0, 22: 0}
    for tag, v in bw.parse_records(np.load(os.path.join(G, "kat_ksw.npz"))["words"]):
        v = [int(x) for x in v]
        qlen, tlen, w, h0, zdrop, bonus, o_del, e_del, o_ins, e_ins = v[:10]
        q = (C.c_uint8 * qlen)(*v[10:10 + qlen])
        t = (C.c_uint8 * tlen)(*v[10 + qlen:10 + qlen + tlen])
        res = v[10 + qlen + tlen:]
        if tag == 20:
            o = [C.c_int() for _ in range(5)]
            sc = lib.ora_ksw_extend2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, w, bonus, zdrop, h0, *[C.byref(x) for x in o])
            assert [sc] + [x.value for x in o] == res
        elif tag == 21:
            n_cigar, cigar = C.c_int(), C.POINTER(C.c_uint32)()
            sc = lib.ora_ksw_global2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, w, C.byref(n_cigar), C.byref(cigar))
            assert [sc, n_cigar.value] + [cigar[i] for i in range(n_cigar.value)] == res
        elif tag == 22:
            xtra = 0x40000 | 0x80000 | (0x10000 if qlen < 250 else 0) | 19
            r = lib.ora_ksw_align2(qlen, q, tlen, t, 5, mat, o_del, e_del, o_ins, e_ins, xtra)
            assert [r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb] == res
        counts[tag] += 1
    assert all(c >= 100 for c in counts.values())


def test_oracle_option_sweep_equals_reference_sam(gold):
    """The restatement under non-default options vs SAM the reference produced with the same options (opt_*.sam.gz),
    incl. -W on 600-700 base reads where mem_flt_chained_seeds (bwamem.c:605) really drops seeds."""
    open(os.path.join(gold["dir"], "long.fq"), "wb").write(gzip.open(os.path.join(G, "long.fq.gz")).read())
    for name in common.GOLDEN_OPTION_SETS:
        fqs = ["pe_1.fq", "pe_2.fq"] if name in common.PE_OPTION_SETS else ["long.fq"] if name in common.LONG_OPTION_SETS else ["se.fq"]
        want = gzip.open(os.path.join(G, f"opt_{name}.sam.gz")).read()
        got = _run_oracle(["mem", "-t", "4"] + common.option_flags(name) + [gold["prefix"]] + [os.path.join(gold["dir"], f) for f in fqs])
        assert got == want, name


def test_oracle_ksw_align2_other_matrices(built):
    """ora_ksw_align2 vs the reference's ksw_align2 under other matrices / gap costs / xtra (kat_ksw_align.npz)."""
    ora = C.CDLL(os.path.join(common.ROOT, "oracle", "liboracle.so"))

    class R(C.Structure):
        _fields_ = [(k, C.c_int) for k in ("score", "te", "qe", "score2", "te2", "tb", "qb")]
    ora.ora_ksw_align2.restype = R
    ora.ora_ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5
    n = 0
    for tag, v in bw.parse_records(np.load(os.path.join(G, "kat_ksw_align.npz"))["words"]):
        if tag != 23:
            continue
        v = [int(x) for x in v]
        qlen, tlen, xtra = v[0], v[1], v[2]
        mat = np.array(v[7:32], dtype=np.int8)
        q = np.array(v[32:32 + qlen], dtype=np.uint8)
        t = np.array(v[32 + qlen:32 + qlen + tlen], dtype=np.uint8)
        r = ora.ora_ksw_align2(qlen, q.ctypes.data, tlen, t.ctypes.data, 5, mat.ctypes.data, v[3], v[4], v[5], v[6], xtra)
        assert [r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb] == v[32 + qlen + tlen:], (n, v[:7])
        n += 1
    assert n >= 300
