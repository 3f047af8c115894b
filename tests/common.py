"""Shared helpers for the test-suite."""
import importlib.util
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle", "bwa_oracle")        # C restatement (checker)
BWAREF = os.path.join(ROOT, "oracle", "_ref", "bwaref")    # reference's own sources (checker, when built)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load_bwahip():
    spec = importlib.util.spec_from_file_location("bwahip", os.path.join(ROOT, "bwa-mem-gpu_amd", "bwahip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


bw = _load_bwahip()


def have_ref():
    return os.path.exists(BWAREF) and os.access(BWAREF, os.X_OK)


def oracle_stages(prefix, fq, out):
    subprocess.check_call([ORACLE, "stages", prefix, fq, out])
    return bw.read_record_file(out)


def by_read(records):
    """[(tag, arr)] -> list of {tag: arr} per read (records following each TAG_READ)."""
    reads = []
    for tag, arr in records:
        if tag == bw.TAG_READ:
            reads.append({})
        else:
            reads[-1][tag] = arr
    return reads


def assert_stage_equal(got, want, tag, what):
    assert len(got) == len(want), f"{what}: {len(got)} reads vs {len(want)}"
    bad = [i for i in range(len(want)) if not np.array_equal(got[i][tag], want[i][tag])]
    assert not bad, f"{what}: {len(bad)} of {len(want)} reads differ, first read {bad[0]}:\n got  {got[bad[0]][tag][:40]}\n want {want[bad[0]][tag][:40]}"


# ---------------------------------------------------------------------------------------------- option sweep
# `bwa mem` option strings used by the option-sweep parity tests.  The checkers (bwa_oracle / bwaref) take them on their
# command line; the product takes a bwahip_opt_t, which opt_from_cli() fills the way main_mem does (fastmap.c:77-175,
# update_a fastmap.c:43-57: -A scales b, T, gap, clip, zdrop and unpaired penalties that were not given explicitly).
OPTION_SETS = {
    "W1": ["-W", "1"], "W5": ["-W", "5"], "W12": ["-W", "12"],
    "k15": ["-k", "15"], "k25": ["-k", "25"],
    "w20": ["-w", "20"], "w300": ["-w", "300"],
    "r1.0": ["-r", "1.0"], "r2.5": ["-r", "2.5"],
    "c50": ["-c", "50"], "c1000": ["-c", "1000"],
    "D0.3": ["-D", "0.3"], "D0.8": ["-D", "0.8"],
    "A2B3": ["-A", "2", "-B", "3"], "A3": ["-A", "3"],
    "O4,8_E2,1": ["-O", "4,8", "-E", "2,1"],
    "L0,10": ["-L", "0,10"], "T20": ["-T", "20"],
    "y5": ["-y", "5"], "s3": ["-s", "3"], "G500": ["-G", "500"], "N2": ["-N", "2"], "d30": ["-d", "30"],
    "X0.3": ["-X", "0.3"], "Q0": ["-Q", "0"], "h2,20": ["-h", "2,20"],
    "M_Y_5": ["-M", "-Y", "-5"], "a_q": ["-a", "-q"],
    "W3_k17_w50": ["-W", "3", "-k", "17", "-w", "50"],
}
# read sets of 600-700 bases: there 0.05*l >= 1.1*W holds for W up to 31 while min_HSP_score = 1.1*W exceeds the score of a
# bare 19..30-base seed, so mem_flt_chained_seeds (bwamem.c:605) really drops seeds (on 150 bp reads it can only re-score them)
LONG_OPTION_SETS = {"W20_long": ["-W", "20"], "W28_long": ["-W", "28"], "W0_long": [], "W25_A2_long": ["-W", "25", "-A", "2"]}
PE_OPTION_SETS = {
    "m5": ["-m", "5"], "U5": ["-U", "5"], "S": ["-S"], "P": ["-P"], "I400,60": ["-I", "400,60"],
    "W2_PE": ["-W", "2"], "A2_PE": ["-A", "2"],
}


# option sets for which tests/golden holds SAM the REFERENCE produced (make_golden.py extras)
GOLDEN_OPTION_SETS = ["W1", "W5", "k15", "w20", "A2B3", "O4,8_E2,1", "W3_k17_w50", "W2_PE", "m5", "W20_long", "W28_long", "W0_long"]


def option_flags(name):
    for d in (OPTION_SETS, PE_OPTION_SETS, LONG_OPTION_SETS):
        if name in d:
            return d[name]
    raise KeyError(name)


def opt_from_cli(args):
    """[flags] -> (bwahip Opt, pes0 or None)."""
    import math
    o = bw.default_opt()
    given = set()
    pes0 = None

    def pair(v):
        p = v.replace(";", ",").split(",")
        return int(p[0]), int(p[1]) if len(p) > 1 else int(p[0])
    ints = {"k": "min_seed_len", "w": "w", "A": "a", "B": "b", "T": "T", "U": "pen_unpaired", "c": "max_occ", "d": "zdrop",
            "m": "max_matesw", "s": "split_width", "G": "max_chain_gap", "N": "max_chain_extend", "W": "min_chain_weight",
            "y": "max_mem_intv", "t": "n_threads"}
    floats = {"r": "split_factor", "D": "drop_ratio", "X": "mask_level"}
    flags = {"P": 0x4, "a": 0x8, "M": 0x10, "S": 0x20, "Y": 0x200, "V": 0x100, "5": 0x800 | 0x1000, "q": 0x1000, "u": 0x2000}
    i = 0
    while i < len(args):
        c = args[i][1]
        if c in flags:
            o.flag |= flags[c]
            i += 1
            continue
        v = args[i + 1]
        i += 2
        if c in ints:
            setattr(o, ints[c], int(v)); given.add(ints[c])
        elif c in floats:
            setattr(o, floats[c], float(v)); given.add(floats[c])
        elif c == "O":
            o.o_del, o.o_ins = pair(v); given |= {"o_del", "o_ins"}
        elif c == "E":
            o.e_del, o.e_ins = pair(v); given |= {"e_del", "e_ins"}
        elif c == "L":
            o.pen_clip5, o.pen_clip3 = pair(v); given |= {"pen_clip5", "pen_clip3"}
        elif c == "h":
            o.max_XA_hits, o.max_XA_hits_alt = pair(v)
        elif c == "Q":
            o.mapQ_coef_len = float(int(v))
            o.mapQ_coef_fac = int(math.log(o.mapQ_coef_len)) if o.mapQ_coef_len > 0 else 0
        elif c == "I":
            p = [float(x) for x in v.split(",")]
            pes0 = (bw.PeStat * 4)()
            for k in range(4):
                pes0[k].failed = 1
            r = pes0[1]
            r.failed, r.avg = 0, p[0]
            r.std = p[1] if len(p) > 1 else p[0] * .1
            r.high = int(p[2] + .499) if len(p) > 2 else int(r.avg + 4. * r.std + .499)
            r.low = int(p[3] + .499) if len(p) > 3 else max(1, int(r.avg - 4. * r.std + .499))
        else:
            raise ValueError(f"opt_from_cli: unsupported option -{c}")
    if "a" in given:
        for f in ("b", "T", "o_del", "e_del", "o_ins", "e_ins", "zdrop", "pen_clip5", "pen_clip3", "pen_unpaired"):
            if f not in given:
                setattr(o, f, getattr(o, f) * o.a)
    bw.lib().bwahip_opt_fill_scmat(bw.C.byref(o))
    return o, pes0
