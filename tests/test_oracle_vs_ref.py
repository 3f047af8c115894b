"""CPU tests run only where the reference build exists (oracle/_ref/bwaref, compiled from the reference's own
sources): randomized SE / PE / ALT / -a runs, oracle vs reference, byte-identical SAM and stage dumps."""
import os
import subprocess
import pytest
import common
from common import bw

pytestmark = pytest.mark.skipif(not common.have_ref(), reason="oracle/_ref/bwaref not built (reference absent)")


def _sam(exe, args):
    return subprocess.run([exe, "mem", "-t", "8"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout


@pytest.mark.parametrize("seed,n,length,sub,indel,nn,chim", [(401, 4000, 150, 10000, 2000, 500, 20000), (402, 1500, 250, 50000, 3000, 500, 30000),
                                                             (403, 2000, 60, 30000, 2000, 3000, 0)])
def test_se_sam_and_stages(small_index, tmp_path, seed, n, length, sub, indel, nn, chim):
    fq = str(tmp_path / "r.fq")
    bw.make_reads(small_index["fa"], fq, None, n, length, sub, indel, nn, seed, chim)
    assert _sam(common.ORACLE, [small_index["prefix"], fq]) == _sam(common.BWAREF, [small_index["prefix"], fq])
    assert _sam(common.ORACLE, ["-a", small_index["prefix"], fq]) == _sam(common.BWAREF, ["-a", small_index["prefix"], fq])
    subprocess.check_call([common.ORACLE, "stages", small_index["prefix"], fq, str(tmp_path / "o.bin")])
    subprocess.check_call([common.BWAREF, "stages", small_index["prefix"], fq, str(tmp_path / "r.bin")])
    assert open(tmp_path / "o.bin", "rb").read() == open(tmp_path / "r.bin", "rb").read()


@pytest.mark.parametrize("seed,length,sub,indel", [(411, 150, 10000, 1000), (412, 100, 60000, 5000)])
def test_pe_sam_incl_alt(small_index, tmp_path, seed, length, sub, indel):
    f1, f2 = str(tmp_path / "1.fq"), str(tmp_path / "2.fq")
    bw.make_reads(small_index["fa"], f1, f2, 8000, length, sub, indel, 1000, seed)
    assert _sam(common.ORACLE, [small_index["prefix"], f1, f2]) == _sam(common.BWAREF, [small_index["prefix"], f1, f2])
    assert _sam(common.ORACLE, ["-K", "100000", small_index["prefix"], f1, f2]) == _sam(common.BWAREF, ["-K", "100000", small_index["prefix"], f1, f2])
    alt = str(tmp_path / "alt")
    for e in ("amb", "ann", "bwt", "pac", "sa"):
        os.symlink(small_index["prefix"] + "." + e, alt + "." + e)
    open(alt + ".alt", "w").write("ctg3\n")
    assert _sam(common.ORACLE, [alt, f1, f2]) == _sam(common.BWAREF, [alt, f1, f2])


def test_mkindex_equals_reference_index(small_index, tmp_path):
    ref_prefix = str(tmp_path / "ref")
    subprocess.check_call([common.BWAREF, "index", small_index["fa"], ref_prefix], stderr=subprocess.DEVNULL)
    for e in ("pac", "ann", "amb", "bwt", "sa"):
        assert open(ref_prefix + "." + e, "rb").read() == open(small_index["prefix"] + "." + e, "rb").read(), e


@pytest.fixture(scope="module")
def sweep_reads(small_index, tmp_path_factory):
    d = tmp_path_factory.mktemp("sweep")
    fq = str(d / "se.fq")
    parts = []
    for k, (n, ln, sub, indel, nn, chim) in enumerate([(700, 150, 15000, 3000, 500, 30000), (250, 250, 50000, 4000, 500, 30000), (100, 650, 40000, 3000, 500, 50000)]):
        p = str(d / f"p{k}.fq")
        bw.make_reads(small_index["fa"], p, None, n, ln, sub, indel, nn, 430 + k, chim)
        parts.append(open(p).read().replace("@r", f"@s{k}_"))
    open(fq, "w").write("".join(parts))
    f1, f2 = str(d / "1.fq"), str(d / "2.fq")
    bw.make_reads(small_index["fa"], f1, f2, 2000, 150, 20000, 2000, 500, 440)
    return fq, f1, f2


@pytest.mark.parametrize("name", sorted(common.OPTION_SETS) + sorted(common.LONG_OPTION_SETS))
def test_option_sweep_se(small_index, sweep_reads, tmp_path, name):
    """Oracle vs the reference under non-default options: SAM and every stage dump."""
    flags = common.option_flags(name)
    fq = sweep_reads[0]
    assert _sam(common.ORACLE, flags + [small_index["prefix"], fq]) == _sam(common.BWAREF, flags + [small_index["prefix"], fq])
    subprocess.check_call([common.ORACLE, "stages", *flags, small_index["prefix"], fq, str(tmp_path / "o.bin")])
    subprocess.check_call([common.BWAREF, "stages", *flags, small_index["prefix"], fq, str(tmp_path / "r.bin")])
    assert open(tmp_path / "o.bin", "rb").read() == open(tmp_path / "r.bin", "rb").read()


@pytest.mark.parametrize("name", sorted(common.PE_OPTION_SETS))
def test_option_sweep_pe(small_index, sweep_reads, name):
    flags = common.option_flags(name)
    _, f1, f2 = sweep_reads
    assert _sam(common.ORACLE, flags + [small_index["prefix"], f1, f2]) == _sam(common.BWAREF, flags + [small_index["prefix"], f1, f2])
