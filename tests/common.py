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
