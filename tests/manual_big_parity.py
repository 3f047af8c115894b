#!/usr/bin/env python3
"""Manual (not collected by pytest) large parity run on the GPU box: reads WITH indels, N's and chimeras (tools/simgen) against a repeat-rich genome,
single-end and paired-end, several read lengths / error rates; bwahip_process_seqs vs the CPU path (oracle/_ref/bwaref if present,
else the C restatement), byte for byte.   python tests/manual_big_parity.py [genome_mbp] [reads] [default|human-like]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
bw = entry.load_bwahip(); bw.lib()
import common
import tools_py as tp

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
profile = sys.argv[3] if len(sys.argv) > 3 else "default"
d = "/dev/shm/big_parity"; os.makedirs(d, exist_ok=True)
prefix = f"{d}/g{mbp}" + ("h" if profile != "default" else "")
lens = tp.contig_lengths(mbp * 1000000)
if not os.path.exists(prefix + ".sa"):
    tp.write_fasta(prefix + ".fa", tp.make_genome(38, lens, repeats=True, profile=profile), lens)
    bw.make_index(prefix + ".fa", prefix)
exe = common.BWAREF if common.have_ref() else common.ORACLE
body = lambda s: b"\n".join(l for l in s.split(b"\n") if not l.startswith(b"@"))
ok = True
with bw.Context(prefix) as ctx:
    #            length sub_ppm indel_ppm n_ppm chim_ppm paired
    for tag, ln, sub, indel, nn, chim, pe in [("pe150", 150, 10000, 3000, 300, 20000, True), ("pe250_noisy", 250, 40000, 8000, 500, 30000, True),
                                              ("se100", 100, 20000, 5000, 1000, 20000, False), ("se400", 400, 30000, 6000, 300, 50000, False)]:
        f1, f2 = f"{d}/{tag}_1.fq", f"{d}/{tag}_2.fq" if pe else None
        bw.make_reads(prefix + ".fa", f1, f2, n, ln, sub, indel, nn, 900 + ln, chim)
        files = [f1, f2] if pe else [f1]
        t0 = time.time()
        want = subprocess.run([exe, "mem", "-t", "16", "-K", "30000000", prefix, *files], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        t_cpu = time.time() - t0
        opt = bw.default_opt(); opt.n_threads = 16
        if pe: opt.flag |= 0x2
        got, n_done = [], 0
        t0 = time.time()
        with bw.FastqReader(*files) as rd:
            while True:
                arr, k = rd.next(30000000)
                if k == 0: break
                got.append(ctx.process_seqs_text_array(arr, k, opt, n_processed=n_done)); n_done += k
        t_gpu = time.time() - t0
        same = b"".join(got) == body(want)
        ok &= same
        print(f"{tag}: {n_done} reads, identical={same}, cpu {t_cpu:.1f}s, gpu path incl. file reading {t_gpu:.1f}s", flush=True)
        if not same:
            g, w = b"".join(got).split(b"\n"), body(want).split(b"\n")
            for i, (x, y) in enumerate(zip(g, w)):
                if x != y:
                    print("first difference at line", i, "\n got ", x[:300], "\n want", y[:300]); break
sys.exit(0 if ok else 1)
