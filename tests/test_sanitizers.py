"""CPU builds under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool):
  * the product's host-side code: csrc/index_io.cpp + csrc/host_final.cpp + csrc/fastq_reader.cpp (driven by tests/san_host_driver.cpp, where the CPU
    oracle stands in for the one GPU call those files make) and tools/mkindex.cpp;
  * the oracle itself (oracle/ora_*.c + main_oracle.c).
Every run must finish without a sanitizer report (-fno-sanitize-recover: any report aborts) AND reproduce the golden
vectors the reference produced (tests/golden)."""
import gzip
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
import pytest
import common

G = common.GOLDEN
ROOT = common.ROOT
CSRC = os.path.join(ROOT, "bwa-mem-gpu_amd", "csrc")
SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def san(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("san"))
    jobs = []
    ora = sorted(f for f in os.listdir(os.path.join(ROOT, "oracle")) if f.startswith("ora_") and f.endswith(".c"))
    for f in ora + ["main_oracle.c"]:
        jobs.append(["gcc", *SAN, "-c", os.path.join(ROOT, "oracle", f), "-o", os.path.join(d, f[:-2] + ".o")])
    hipinc = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-std=c++17", "-Wno-unused-result"]   # hip_runtime.h as plain host C++: types only
    for src, obj in ((os.path.join(CSRC, "index_io.cpp"), "index_io.o"), (os.path.join(CSRC, "host_final.cpp"), "host_final.o"), (os.path.join(CSRC, "fastq_reader.cpp"), "fastq_reader.o"),
                     (os.path.join(ROOT, "tests", "san_host_driver.cpp"), "driver.o")):
        jobs.append(["g++", *SAN, *hipinc, "-c", src, "-o", os.path.join(d, obj)])
    jobs.append(["g++", *SAN, "-std=c++17", "-fopenmp", "-o", os.path.join(d, "mkindex"), os.path.join(ROOT, "bwa-mem-gpu_amd", "tools", "mkindex.cpp")])
    with ThreadPoolExecutor(4) as ex:
        for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
            assert r.returncode == 0, r.stderr[-2000:]
    oo = [os.path.join(d, f[:-2] + ".o") for f in ora]
    link = ["-fsanitize=address,undefined", "-lm", "-lz", "-lpthread"]
    subprocess.check_call(["gcc", "-o", os.path.join(d, "bwa_oracle"), os.path.join(d, "main_oracle.o"), *oo, *link])
    subprocess.check_call(["g++", "-o", os.path.join(d, "host_driver"), os.path.join(d, "driver.o"), os.path.join(d, "index_io.o"),
                           os.path.join(d, "host_final.o"), os.path.join(d, "fastq_reader.o"), *oo, *link])
    fa = os.path.join(d, "g60k.fa")
    open(fa, "wb").write(gzip.open(os.path.join(G, "g60k.fa.gz")).read())
    open(os.path.join(d, "g60k.alt"), "wb").write(open(os.path.join(G, "g60k.alt"), "rb").read())
    for n in ("se.fq", "pe_1.fq", "pe_2.fq"):
        open(os.path.join(d, n), "wb").write(gzip.open(os.path.join(G, n + ".gz")).read())
    return d


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=ENV)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode(errors="replace")[-3000:]
    return r.stdout


def test_mkindex_clean_and_identical_to_reference_index(san):
    _run([os.path.join(san, "mkindex"), os.path.join(san, "g60k.fa"), os.path.join(san, "g60k")])
    want = dict(reversed(l.split()) for l in open(os.path.join(G, "index.sha256")))
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        assert hashlib.sha256(open(os.path.join(san, "g60k." + ext), "rb").read()).hexdigest() == want["g60k." + ext]


@pytest.mark.parametrize("name,extra,fqs", [("se", [], ["se.fq"]), ("se_all", ["-a"], ["se.fq"]), ("pe", [], ["pe_1.fq", "pe_2.fq"])])
def test_host_finalisation_clean_and_identical_to_reference_sam(san, name, extra, fqs):
    """index_io.cpp + host_final.cpp (the gpu_final = 0 / gpu_pair = 0 path of bwahip_process_seqs) under ASan/UBSan; SE in two
    batches with their true n_processed."""
    test_mkindex_clean_and_identical_to_reference_index(san)
    got = _run([os.path.join(san, "host_driver"), *extra, os.path.join(san, "g60k"), *[os.path.join(san, f) for f in fqs]])
    assert got == gzip.open(os.path.join(G, name + ".sam.gz")).read()


@pytest.mark.parametrize("name,extra,fqs", [("se", ["-t", "3", "-K", "40000"], ["se.fq"]), ("pe", ["-t", "2"], ["pe_1.fq", "pe_2.fq"])])
def test_oracle_clean_and_identical_to_reference_sam(san, name, extra, fqs):
    test_mkindex_clean_and_identical_to_reference_index(san)
    got = _run([os.path.join(san, "bwa_oracle"), "mem", *extra, os.path.join(san, "g60k"), *[os.path.join(san, f) for f in fqs]])
    assert got == gzip.open(os.path.join(G, name + ".sam.gz")).read()


@pytest.mark.parametrize("want,fqs,chunk", [("cases_1.se.txt", ["cases_1.fq"], 100000), ("cases.pe.txt", ["cases_1.fq", "cases_2.fq"], 100000),
                                            ("pairs.pe3000.txt", ["pairs_1.fq", "pairs_2.fq.gz"], 3000), ("pairs.se777.txt", ["pairs_1.fq"], 777)])
def test_fastq_reader_clean_and_identical_to_reference_bseq_read(san, want, fqs, chunk):
    """csrc/fastq_reader.cpp under ASan/UBSan on the hand-made FASTA/FASTQ cases (CRLF, multi-line, truncated last record, fewer mates,
    gzip, -K chunking) against the batches the reference's bseq_read cut (tests/golden/fastq)."""
    F = os.path.join(G, "fastq")
    got = _run([os.path.join(san, "host_driver"), "-d", str(chunk), *[os.path.join(F, f) for f in fqs]])
    assert got == open(os.path.join(F, want), "rb").read()
