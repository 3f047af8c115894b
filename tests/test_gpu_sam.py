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


def _first_diff(got, want):
    g, w = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(g, w)):
        if a != b:
            return f"SAM differs at line {i} of {len(w)}:\n got  {a[:300]}\n want {b[:300]}"
    return f"SAM line counts differ: {len(g)} vs {len(w)}"
