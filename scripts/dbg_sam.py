"""debug: run one SE read set through bwahip_process_seqs and the oracle; save both SAMs under gpurun_out/"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from common import bw
import __graft_entry__ as g
g.build()
d = "/tmp/dbg_sam"; os.makedirs(d, exist_ok=True)
fa = f"{d}/g1.fa"
bw.make_genome(fa, 11, [600000, 300000, 100000], repeats=True)
bw.make_index(fa, f"{d}/g1")
n, length, sub, indel, nn, seed, chim = [int(x) for x in sys.argv[1:8]] if len(sys.argv) > 7 else (1500, 250, 50000, 3000, 500, 205, 30000)
fq = f"{d}/r.fq"
bw.make_reads(fa, fq, None, n, length, sub, indel, nn, seed, chim)
names, seqs, quals = bw.read_fastq(fq)
want = subprocess.run([common.ORACLE, "mem", "-t", "8", f"{d}/g1", fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
with bw.Context(f"{d}/g1") as ctx:
    opt = bw.default_opt(); opt.n_threads = 8
    fq0 = f"{d}/r0.fq"
    bw.make_reads(fa, fq0, None, 4000, 150, 10000, 2000, 500, 201, 20000)
    n0, s0, q0 = bw.read_fastq(fq0)
    first = b"".join(ctx.process_seqs(n0, s0, q0, opt))
    w0 = subprocess.run([common.ORACLE, "mem", "-t", "8", f"{d}/g1", fq0], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    print("first batch equal:", first == w0)
    got = b"".join(ctx.process_seqs(names, seqs, quals, opt))
    ctx.tune(gpu_final=0)
    got_host = b"".join(ctx.process_seqs(names, seqs, quals, opt))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "dbg_got.sam"), "wb").write(got)
open(os.path.join(ROOT, "gpurun_out", "dbg_want.sam"), "wb").write(want)
print("equal:", got == want, "host equal:", got_host == want, len(got), len(want))
g_, w_ = got.split(b"\n"), want.split(b"\n")
nd = 0
for i, (a, b) in enumerate(zip(g_, w_)):
    if a != b:
        print("line", i, "\n got ", a[:400], "\n want", b[:400]); nd += 1
        if nd > 5: break
