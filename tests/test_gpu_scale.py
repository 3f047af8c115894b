"""Parity on the bench workload itself: a repeat-rich genome of the bench generator (seed 38, repeat families up to
1 400 copies) with default thresholds, so that the heavy-read hand-offs (k_smem_heavy, k_chain_big, k_chain_flt,
k_extend_spec, rank-sort dedup) are taken by the reads that really need them, not forced.  HIP path vs the oracle,
bit-exact at the interval, filtered-chain and region boundaries, plus SAM for a slice."""
import os
import numpy as np
import pytest
import common
from common import bw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench_genome(built, tmp_path_factory):
    import tools_py as tp
    d = tmp_path_factory.mktemp("scale")
    mbp = int(os.environ.get("BWAHIP_SCALE_TEST_MBP", "24"))
    lens = tp.contig_lengths(mbp * 1000000)
    genome = tp.make_genome(38, lens, repeats=True)
    prefix = str(d / "g")
    tp.write_fasta(prefix + ".fa", genome, lens)
    bw.make_index(prefix + ".fa", prefix)
    return {"prefix": prefix, "genome": genome, "lens": lens, "dir": str(d)}


def test_bench_workload_matches_oracle(bench_genome, tmp_path):
    import tools_py as tp
    n = int(os.environ.get("BWAHIP_SCALE_TEST_READS", "30000"))
    reads = tp.make_reads(bench_genome["genome"], bench_genome["lens"], n, 150, sub_ppm=10000, seed=102)
    fq = str(tmp_path / "r.fq")
    tp.write_fastq(fq, reads)
    _, seqs, _ = bw.read_fastq(fq)
    want = common.by_read(common.oracle_stages(bench_genome["prefix"], fq, str(tmp_path / "o.bin")))
    codes, off = bw.pack_reads(seqs)
    with bw.Context(bench_genome["prefix"]) as ctx:
        got = common.by_read(ctx.run_stages(codes, off, [bw.STAGE_INTV, bw.STAGE_CHAIN_FLT, bw.STAGE_REGS_PRE, bw.STAGE_REGS]))
        cnt = ctx.counters()
    for st, what in [(bw.STAGE_INTV, "intervals"), (bw.STAGE_CHAIN_FLT, "filtered chains"), (bw.STAGE_REGS_PRE, "regions before dedup"),
                     (bw.STAGE_REGS, "regions")]:
        common.assert_stage_equal(got, want, st, f"{what}[bench genome]")
    # the point of this test: the hand-off kernels ran on their own account
    assert cnt["heavy_intv"] > 0, "no read was handed to k_smem_heavy"
    assert cnt["max_seeds"] > 512, "no read reached k_chain_big's threshold"
    assert cnt["max_chains"] >= 16, "no read reached k_extend_spec's threshold"


def test_bench_workload_sam_identical_to_reference(bench_genome, tmp_path):
    """End to end on the same workload: bwahip_process_seqs vs the reference's own mem_process_seqs (oracle/_ref/bwaref
    when present, else the C restatement), SE and PE, byte-identical SAM."""
    import subprocess
    import tools_py as tp
    exe = common.BWAREF if common.have_ref() else common.ORACLE
    n = int(os.environ.get("BWAHIP_SCALE_TEST_SAM_READS", "20000"))
    opt = bw.default_opt()
    opt.n_threads = 8
    # SE
    reads = tp.make_reads(bench_genome["genome"], bench_genome["lens"], n, 150, sub_ppm=10000, seed=131)
    fq = str(tmp_path / "se.fq")
    tp.write_fastq(fq, reads)
    names, seqs, quals = bw.read_fastq(fq)
    want = subprocess.run([exe, "mem", "-t", "8", bench_genome["prefix"], fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    with bw.Context(bench_genome["prefix"]) as ctx:
        got = b"".join(ctx.process_seqs(names, seqs, quals, opt))
        body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
        assert body(got) == body(want), "SE SAM differs from the CPU path on the bench workload"
        # PE (configs[2] shape: 2x150, insert N(500,50^2), FR): insert-size statistics, mate rescue, pairing, paired SAM on the GPU
        reads = tp.make_reads(bench_genome["genome"], bench_genome["lens"], n, 150, sub_ppm=10000, seed=133, paired=True)
        fqs = tp.write_fastq_fixed(str(tmp_path / "pe"), reads, True)
        names_pe = tp.fixed_names(n, True)
        opt.flag |= 0x2
        want = subprocess.run([exe, "mem", "-t", "8", bench_genome["prefix"], *fqs], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        secs, got = tp.process_seqs_bulk(bw, ctx, opt, names_pe, reads)
        assert body(got) == body(want), "PE SAM differs from the CPU path on the bench workload"
        assert ctx.last_pe_stats()[0][1]["failed"] == 0
