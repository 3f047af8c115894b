#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X BWA-MEM path on BASELINE.json's headline configuration: 150 bp paired-end reads (2x150, insert
N(500,50^2), FR) against an hg38-scale genome (3.1 Gbp; GRCh38 itself is not available offline, so a deterministic synthetic genome
with repeat families is used and named as such).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the whole of mem_process_seqs -- hot path (mem_align1_core), insert-size statistics, mate rescue, pairing,
mark-primary, CIGAR/NM/MD/mapQ, SAM text -- over this rank's read set (default 10 M reads = 5 M pairs, BASELINE configs[2]) in batches of
1 M reads, exactly as `bwa mem -K 150000000` cuts them.  Every step is measured three ways, each region bracketed by barrier + synchronize:

  value             BASELINE.md section 3's metric: FASTQ files in -> SAM text out ("first batch in -> last SAM out"), through the product's own
                    batch driver behind the C ABI (bwahip_stream_run: the library's parallel FASTQ reader, --overlap contexts per GPU taking
                    whole batches, the writer thread emitting the SAM in input order).  Input files on tmpfs, output to /dev/null in the timed
                    steps; one untimed pass writes the SAM to a file, which is compared byte for byte with the reference CPU path's output.
  gpu_pipeline      the same batches with the reads (codes, names, qualities) already resident in HBM and the SAM text left in HBM, double
                    buffered on --overlap contexts: the GPU side alone.
  single_context    the same, one batch at a time on one context: the region the per-kernel durations (`kernel_ms`) and the `roofline` of the
                    BWT-search kernel are measured in.

`other_configs` adds BASELINE configs[1] (1 M x 150 bp SE, 1 %) and configs[4] (1 M x 250 bp SE, 5 %) on the same 3.1 Gbp index, each with its
single-context rate, per-kernel durations and an in-run SAM parity check against the reference CPU path.  Reads are sharded per rank with no
data-path collective ("weak" scaling); RCCL is used once, to broadcast the index from rank 0.  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def usable_cpus():
    """CPUs this process can really use: affinity mask and the container's CFS quota (cgroup v2 cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def config_label(pe, rl, sub_ppm):
    if pe and rl == 150 and sub_ppm == 10000:
        return "BASELINE configs[2] shape"
    if not pe and rl == 150 and sub_ppm == 10000:
        return "BASELINE configs[1] shape"
    if not pe and rl == 250 and sub_ppm == 50000:
        return "BASELINE configs[4] shape"
    return "not a BASELINE configuration"


class Resident:
    """A read set resident in HBM (codes, names, qualities) and the runs over it, batch by batch."""

    def __init__(self, bw, tp, torch, dev, reads, pe, batch):
        self.bw, self.torch, self.pe = bw, torch, pe
        self.n, self.rl = reads.shape
        self.batch = min(batch, self.n)
        self.n_batches = (self.n + self.batch - 1) // self.batch
        rl = self.rl
        self.codes = torch.empty(self.n * rl, dtype=torch.uint8, device=dev)
        for b0 in range(0, self.n, 2000000):                   # staged: the ASCII -> code table look-up doubles the host footprint otherwise
            b1 = min(self.n, b0 + 2000000)
            self.codes[b0 * rl:b1 * rl] = torch.from_numpy(bw.NT4[reads[b0:b1].reshape(-1)]).to(dev)
        self.off = torch.arange(self.batch + 1, dtype=torch.int64, device=dev) * rl
        # the text the SAM stage prints, resident too: names (fixed width, NUL terminated), qualities (all 'I': one batch worth, shared)
        self.names_host = tp.fixed_names(self.n, pe)
        self.nw = self.names_host.shape[1]
        self.names = torch.zeros(self.n * self.nw + 64, dtype=torch.uint8, device=dev)
        self.names[:self.n * self.nw] = torch.from_numpy(self.names_host.reshape(-1)).to(dev)
        self.name_off = torch.arange(self.batch + 1, dtype=torch.int64, device=dev) * self.nw
        self.qual = torch.full((self.batch * rl + 64,), ord("I"), dtype=torch.uint8, device=dev)
        self.qual_off = torch.arange(self.batch, dtype=torch.int64, device=dev) * rl
        torch.cuda.synchronize()

    def bounds(self, b):
        b0 = b * self.batch
        return b0, min(self.n, b0 + self.batch)

    def attach(self, cx, b0, b1):
        cx.batch_attach(b1 - b0, self.codes.data_ptr() + b0 * self.rl, self.off.data_ptr(), self.rl, (b1 - b0) * self.rl)
        cx.batch_attach_text(self.qual.data_ptr(), self.qual_off.data_ptr(), self.names.data_ptr() + b0 * self.nw, self.name_off.data_ptr())

    def step(self, ctxs, opt, collect=None):
        """One pass over the reads.  With several contexts (double buffering) context t takes batches t, t + n, ... on its own streams,
        driven by its own host thread."""
        def work(t):
            for b in range(t, self.n_batches, len(ctxs)):
                b0, b1 = self.bounds(b)
                self.attach(ctxs[t], b0, b1)
                km = ctxs[t].batch_run_sam(opt, n_processed=b0)
                if collect is not None:
                    collect.append(km)
        if len(ctxs) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(t,)) for t in range(len(ctxs))]
            for x in th:
                x.start()
            for x in th:
                x.join()

    def count(self, ctx, opt):
        """Algorithmic work per launch, counted by the kernels themselves: one untimed pass, counters read after every batch.  Also returns
        the SAM of batch 0."""
        cnts, sam0 = [], None
        for b in range(self.n_batches):
            b0, b1 = self.bounds(b)
            self.attach(ctx, b0, b1)
            ctx.batch_run_sam(opt, n_processed=b0)
            cnts.append(ctx.counters())
            if b == 0:
                sam0 = ctx.batch_sam()
        counters = {k: (max(c[k] for c in cnts) if k.endswith("_max") or k.startswith("max_") else sum(c[k] for c in cnts)) for k in cnts[0]}
        return counters, sam0


def per_read(counters, n):
    return {"bwt_extend": round(counters["extend"] / n, 1), "occ_blocks": round(counters["blocks"] / n, 1),
            "sa_lookups": round(counters["sa"] / n, 2), "lf_steps": round(counters["lf"] / n, 1), "dp_cells": round(counters["cells"] / n, 1),
            "dp_rows_1col": round(counters["dp_rows_1col"] / n, 1), "dp_rows_ncol": round(counters["dp_rows_ncol"] / n, 1)}


def tails(counters):
    return {k: round(v / 100.0, 1) for k, v in counters.items() if k.endswith("_max")}


def main():
    # exactly ONE line on stdout: libraries (RCCL prints a version banner) write there too, so fd 1 is pointed at stderr for the
    # whole run and the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("BWAHIP_BENCH_MBP", "3100")),
                    help="size of the synthetic genome (GRCh38 itself is not available offline); hg38 is 3.1 Gbp")
    ap.add_argument("--genome-profile", default=os.environ.get("BWAHIP_BENCH_PROFILE", "default"), choices=["default", "human-like"],
                    help="repeat content of the synthetic genome: default = dispersed families up to 1 400 copies; human-like = 45 %% of the bases in "
                         "Alu/L1-like families of 10^4..10^6 copies at 5-15 %% divergence plus satellite arrays")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BWAHIP_BENCH_READS", "10000000")),
                    help="reads per GPU and step (PE: mates interleaved, so half as many pairs)")
    ap.add_argument("--batch", type=int, default=1000000, help="reads per mem_process_seqs batch (bwa mem -K 150000000 at 150 bp)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--se", action="store_true", help="single-end reads (BASELINE configs[1]) instead of paired-end")
    ap.add_argument("--sub-ppm", type=int, default=10000, help="substitution errors per million bases (configs[4]: 50000)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BWAHIP_BENCH_CPU_READS", "1000000")),
                    help="reads of the same workload given to the CPU path (and compared byte for byte with the GPU path's SAM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file-to-file stream (value falls back to gpu_pipeline)")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("BWAHIP_BENCH_OVERLAP", "2")),
                    help="contexts per GPU taking the batches in turn (double buffering) in gpu_pipeline; 1 = one batch at a time only")
    ap.add_argument("--stream-contexts", type=int, default=int(os.environ.get("BWAHIP_BENCH_STREAM_CONTEXTS", "3")),
                    help="contexts per GPU bwahip_stream_run deals the batches to (file to file: a third context covers the others' host phases)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    pe = not args.se
    if pe:
        args.reads &= ~1
        args.batch &= ~1
    args.batch = min(args.batch, args.reads)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal on a one-GPU box: BWAHIP_BENCH_BACKEND=gloo BWAHIP_BENCH_ONE_DEVICE=1 runs every rank on cuda:0
        backend = os.environ.get("BWAHIP_BENCH_BACKEND", "nccl")
        if os.environ.get("BWAHIP_BENCH_ONE_DEVICE"):
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    import __graft_entry__ as entry
    bw = entry.load_bwahip()
    bw.lib()  # no fallback: fails loudly if libbwahip.so is missing
    import tools_py as tp

    workdir = os.environ.get("BWAHIP_BENCH_DIR", "/dev/shm/bwahip_bench" if os.path.isdir("/dev/shm") else "/tmp/bwahip_bench")
    os.makedirs(workdir, exist_ok=True)
    human = args.genome_profile == "human-like"
    prefix = os.path.join(workdir, f"g{args.genome_mbp}" + ("h" if human else ""))
    fa = prefix + ".fa"
    t_index = 0.0
    lens = tp.contig_lengths(args.genome_mbp * 1000000)
    cpus = usable_cpus()
    # ---------------- genome + index: rank 0 builds (stock bwa format, cached under /dev/shm), the others receive it over RCCL
    genome = None
    if rank == 0:
        t0 = time.time()
        if not (os.path.exists(prefix + ".sa") and os.path.exists(fa)):
            genome = tp.make_genome(38, lens, repeats=True, profile=args.genome_profile)
            tp.write_fasta(fa, genome, lens)
            log(f"genome {args.genome_mbp} Mbp ({args.genome_profile}) generated and written: {time.time() - t0:.1f}s")
            bw.make_index(fa, prefix)
        t_index = time.time() - t0
        log(f"genome + index ready: {t_index:.1f}s")
    if world > 1:
        dist.barrier()
    t0 = time.time()
    holder = None
    index_distribution = "bwahip_init_from_files"
    if world == 1:
        ctx = bw.Context(prefix, local_rank)
    else:
        # rank 0 reads the files, every rank receives the index over RCCL into its own HBM (bwahip_init_rccl, the C ABI's collective);
        # the 128-byte RCCL id travels over the already initialised torch.distributed group
        box = [bw.Context.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        try:
            ctx = bw.Context.from_rccl(prefix if rank == 0 else None, rank, world, box[0], local_rank)
            index_distribution = "bwahip_init_rccl (ncclBroadcast inside libbwahip.so)"
            ok = 1
        except bw.BwahipError as e:
            log(f"rank {rank}: bwahip_init_rccl failed ({e}); falling back to torch.distributed broadcast + bwahip_init_device")
            ok = 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:                                   # any rank failed: everybody takes the torch.distributed path
            if ok:
                ctx.close()
            index_distribution = "torch.distributed broadcast + bwahip_init_device"
            if rank == 0:
                ctx = bw.Context(prefix, local_rank)
            ctx_or_holder = tp.broadcast_index(bw, dist, torch, prefix if rank == 0 else None, rank, local_rank)
            if rank != 0:
                ctx, holder = ctx_or_holder
    t_bcast = time.time() - t0
    log(f"rank {rank}: index resident in HBM ({t_bcast:.1f}s)")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # ---------------- reads: every rank its own set (seed 103 + rank), 1 % substitutions, 50 % reverse strand; PE: FR pairs,
    # insert N(500, 50^2) clipped to [300, 700], mates interleaved.  The genome bytes come from the FASTA rank 0 wrote.
    if genome is None:
        genome = tp.read_fasta_bases(fa, lens)
    t0 = time.time()
    reads = tp.make_reads(genome, lens, args.reads, args.read_len, sub_ppm=args.sub_ppm, seed=(103 if pe else 102) + rank, paired=pe)
    log(f"rank {rank}: {args.reads} reads generated ({time.time() - t0:.1f}s)")
    rl = args.read_len
    res = Resident(bw, tp, torch, dev, reads, pe, args.batch)
    n_batches = res.n_batches
    opt = bw.default_opt()
    if pe:
        opt.flag |= 0x2
    opt.n_threads = max(1, int(os.environ.get("BWAHIP_BENCH_HOST_THREADS", cpus // world)))

    # ---------------- single_context: one batch at a time on one context (kernel_ms, roofline)
    for w in range(args.warmup):
        kms = []
        res.step([ctx], opt, kms)
        log(f"warmup {w}: kernel ms of batch 0 {dict((k, round(v, 2)) for k, v in kms[0].items())}")
    sync_all()
    t0 = time.time()
    kms = []
    for _ in range(args.steps):
        res.step([ctx], opt, kms)
    sync_all()
    elapsed = max_over_ranks(time.time() - t0)
    single = {"value": round(args.reads * world * args.steps / elapsed, 1), "unit": "reads/s", "ms_per_step": round(elapsed / args.steps * 1e3, 3), "steps": args.steps,
              "what": "reads resident in HBM, SAM left in HBM, one batch at a time on one context: the timed region `kernel_ms` and `roofline` are measured in"}
    # ---------------- gpu_pipeline: the same steps double buffered on --overlap contexts sharing the index
    clones = [ctx.clone() for _ in range(max(args.overlap, 1 if args.no_e2e else args.stream_contexts) - 1)]
    ctxs = [ctx] + clones[:max(0, args.overlap - 1)]
    sctxs = [ctx] + clones[:max(0, args.stream_contexts - 1)]
    dbuf = None
    if len(ctxs) > 1 and n_batches > 1:
        for _ in range(max(1, min(args.warmup, 2))):
            res.step(ctxs, opt)
        sync_all()
        t0 = time.time()
        for _ in range(args.steps):
            res.step(ctxs, opt)
        sync_all()
        el2 = max_over_ranks(time.time() - t0)
        dbuf = {"value": round(args.reads * world * args.steps / el2, 1), "unit": "reads/s", "ms_per_step": round(el2 / args.steps * 1e3, 3), "contexts_per_gpu": len(ctxs),
                "what": "reads (codes, names, qualities) resident in HBM when the timed region starts, SAM text left in HBM; the batches of a step are taken in turn by "
                        "several contexts on the GPU (bwahip_ctx_clone: shared index, own streams and host thread each), so that one batch's latency-bound kernels and "
                        "serial tails run under another batch's throughput-bound kernels"}
        log(f"gpu pipeline, double buffered ({len(ctxs)} contexts): {dbuf['value']:.0f} reads/s vs {single['value']:.0f} one batch at a time")
    counters, sam_resident0 = res.count(ctx, opt)
    sam_bytes = len(sam_resident0)
    if pe:
        log(f"batch 0: insert size FR {ctx.last_pe_stats()[0][1]}; mate rescue {ctx.pe_counters}")

    # ---------------- value: FASTQ files -> SAM text through the library's batch driver (bwahip_stream_run), this rank's own files
    stream = None
    stream_sam_path = os.path.join(workdir, f"stream_r{rank}.sam")
    K_bases = args.batch * rl
    if not args.no_e2e:
        t0 = time.time()
        fqs = tp.write_fastq_fixed(os.path.join(workdir, f"bench_r{rank}"), reads, pe)
        log(f"rank {rank}: FASTQ files written ({time.time() - t0:.1f}s): {' '.join(fqs)}")
        fq2 = fqs[1] if pe else None
        reader_threads = int(os.environ.get("BWAHIP_BENCH_READER_THREADS", max(2, min(8, opt.n_threads // 2))))
        # untimed pass 0: SAM to a file on tmpfs (kept for the parity check below); buffers grow to the batch size here
        fd = os.open(stream_sam_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            st0 = bw.stream_run(sctxs, fqs[0], fq2, fd, opt, chunk_bases=K_bases, reader_threads=reader_threads)
        finally:
            os.close(fd)
        log(f"stream pass 0 (to a tmpfs file, buffers growing): {st0.n_reads} reads, {st0.sam_bytes} SAM bytes in {st0.seconds:.3f}s")
        devnull = os.open("/dev/null", os.O_WRONLY)
        for _ in range(max(0, args.warmup - 1)):
            bw.stream_run(sctxs, fqs[0], fq2, devnull, opt, chunk_bases=K_bases, reader_threads=reader_threads)
        sync_all()
        t0 = time.time()
        sts = [bw.stream_run(sctxs, fqs[0], fq2, devnull, opt, chunk_bases=K_bases, reader_threads=reader_threads) for _ in range(args.steps)]
        sync_all()
        el3 = max_over_ranks(time.time() - t0)
        os.close(devnull)
        # the same once more with the SAM written to a tmpfs file (one memcpy of 420 bytes per read more, on the writer thread)
        fd = os.open(stream_sam_path + ".2", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            st_file = bw.stream_run(sctxs, fqs[0], fq2, fd, opt, chunk_bases=K_bases, reader_threads=reader_threads)
        finally:
            os.close(fd)
            os.unlink(stream_sam_path + ".2")
        assert all(s.n_reads == args.reads and s.sam_bytes == st0.sam_bytes for s in sts + [st_file])
        stream = {"value": round(args.reads * world * args.steps / el3, 1), "unit": "reads/s", "ms_per_step": round(el3 / args.steps * 1e3, 3),
                  "contexts_per_gpu": len(sctxs), "reader_threads": reader_threads, "host_threads_per_gpu": opt.n_threads,
                  "sam_bytes_per_step": int(st0.sam_bytes), "batches_per_step": int(st0.n_batches),
                  "seconds_inside_driver": [round(s.seconds, 4) for s in sts],
                  "workers_waiting_for_reader_s": round(sum(s.reader_wait_s for s in sts) / args.steps, 4),
                  "workers_in_process_seqs_s": round(sum(s.gpu_busy_s for s in sts) / args.steps, 4),
                  "writer_in_write_s": round(sum(s.write_s for s in sts) / args.steps, 4),
                  "to_tmpfs_file_reads_per_s": round(st_file.n_reads / st_file.seconds, 1),
                  "what": "bwahip_stream_run: two FASTQ files on tmpfs (one for --se) -> the library's parallel reader -> whole -K batches with their true n_processed "
                          "on the contexts -> SAM text in input order to /dev/null (to_tmpfs_file_reads_per_s: to a file); opening the files to the last SAM "
                          "byte, PCIe both ways and all host work included"}
        log(f"file to file: {stream['value']:.0f} reads/s ({el3 / args.steps:.3f}s per step); to a tmpfs file {stream['to_tmpfs_file_reads_per_s']:.0f}")
    for c2 in clones:
        c2.close()

    # ---------------- other BASELINE configurations on the same index (rank 0, N = 1)
    other = {}
    if rank == 0 and world == 1 and not args.no_other_configs and not args.no_cpu_baseline:
        for key, (o_rl, o_sub, o_seed) in (("configs[1]", (150, 10000, 102)), ("configs[4]", (250, 50000, 105))):
            if (not pe) and o_rl == rl and o_sub == args.sub_ppm:
                continue
            other[key] = other_config(bw, tp, torch, dev, ctx, genome, lens, prefix, workdir, cpus, o_rl, o_sub, o_seed, args.genome_mbp)

    if rank == 0:
        n_launch = len(kms)
        if stream:
            value, ms_step, what = stream["value"], stream["ms_per_step"], "file to file (bwahip_stream_run)"
        elif dbuf:
            value, ms_step, what = dbuf["value"], dbuf["ms_per_step"], "gpu_pipeline (reads resident in HBM)"
        else:
            value, ms_step, what = single["value"], single["ms_per_step"], "single_context (reads resident in HBM)"
        k1 = float(np.mean([k["k_smem"] for k in kms]))
        # algorithmic bytes of the BWT-search kernel per launch: 64 B per Occ block touched by bwt_extend +
        # the read bytes in + 32 B per interval out, counted by the kernel itself (the few reads k_smem hands to
        # k_smem_heavy, and pass 3 which runs in k_smem3, are counted by those kernels and subtracted)
        alg_total = (64 * (counters["blocks"] - counters["heavy_blocks"] - counters["pass3_blocks"]) + args.reads * rl +
                     32 * (counters["intv"] - counters["heavy_intv"] - counters["pass3_intv"]))
        alg_bytes = alg_total / n_batches
        achieved = alg_bytes / (k1 * 1e-3) / 1e9
        kind = (f"{args.reads // 2} pairs (2x{rl} bp, FR, insert N(500,50^2) clipped [300,700])" if pe else f"{args.reads} SE reads of {rl} bp")
        gdesc = ("45 % of the bases in Alu/L1-like families of 10^4..10^6 copies at 5-15 % divergence, satellite arrays" if human else "repeat families up to 1 400 copies")
        out = {
            "metric": f"reads/s aligned ({rl} bp {'PE' if pe else 'SE'} vs hg38-scale {args.genome_mbp} Mbp synthetic genome), {what}",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{kind}, {args.sub_ppm / 10000:g}% substitutions, per GPU and step, in batches of {args.batch} reads (-K {args.batch * rl}) vs "
                                   f"{args.genome_mbp} Mbp synthetic genome ({gdesc}; GRCh38 not available offline); {config_label(pe, rl, args.sub_ppm)}",
                       "launch_workload": f"one batch of {args.batch} {'PE' if pe else 'SE'} reads of {rl} bp, {args.sub_ppm / 10000:g}% substitutions vs {args.genome_mbp} Mbp synthetic genome"
                                          + (" (human-like)" if human else ""),
                       "reads_per_gpu": args.reads, "batch_reads": args.batch, "read_len": rl, "paired": pe, "genome_mbp": args.genome_mbp, "genome_profile": args.genome_profile,
                       "stages": ctx.stage_names(),
                       "output": ctx.output_description(pe),
                       "index_in_hbm": ctx.index_footprint(), "index_build_s": round(t_index, 1), "index_to_hbm_s": round(t_bcast, 2), "index_distribution": index_distribution, "host_cpus_usable": cpus},
            "kernel_ms": {k: round(float(np.mean([x[k] for x in kms])), 3) for k in kms[0]},
            "launches_timed": n_launch, "sam_bytes_per_batch": sam_bytes,
            "per_read": per_read(counters, args.reads),
            "tail_us": tails(counters),
            "tail_counts": {"max_extends_per_read": counters.get("max_extends"), "max_seeds_per_read": counters.get("max_seeds"), "max_chains_per_read": counters.get("max_chains")},
            "roofline": {"kernel": "k_smem", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k1, 3), "reads_per_launch": args.batch},
        }
        out["single_context"] = single
        if dbuf:
            out["gpu_pipeline"] = dbuf
            out["sum_kernel_ms_over_gpu_pipeline_ms_per_batch"] = round(sum(out["kernel_ms"].values()) / (dbuf["ms_per_step"] / n_batches), 3)
        if stream:
            out["file_to_file"] = stream
        # the other kernels SURVEY.md 8(d) asks figures for: the SA look-up kernels against the same HBM roofline, the integer DP of the
        # extension in cell updates per second (no roofline fraction: VALU / LDS issue bound, no MFMA work anywhere)
        out["other_kernels"] = other_kernels(out["kernel_ms"], counters, n_batches)
        # HBM traffic of the same kernel from the committed PMC passes of this very command (counters cannot be read from
        # inside the process); quoted only when the per-launch workload (one batch) AND the kernel sources are the ones the
        # counters were collected on
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("launch_workload") == out["config"]["launch_workload"] and pt.get("kernel_src_sha256") == kernel_src_sha():
                out["roofline"]["traffic"] = int(pt["fetch_bytes_per_launch"] + pt["write_bytes_per_launch"])
                out["roofline"]["traffic_source"] = pt["source"]
        except (OSError, ValueError, KeyError):
            pass
        ceil = measured_ceilings() if world == 1 else None
        if ceil:
            # what this very device sustains for the kernel's access pattern (dependent random 64-byte gathers, one lane per
            # block, 3 GB table) and for a plain streaming copy -- SURVEY.md 8(d); `peak` above stays the 8 TB/s spec figure
            out["roofline"]["measured_gather64_GBps"] = ceil["gather64_GBps"]
            out["roofline"]["measured_stream_copy_GBps"] = ceil["stream_copy_GBps"]
            out["roofline"]["frac_of_measured_gather"] = round(achieved / ceil["gather64_GBps"], 4)
        parity_ok = None
        if not args.no_cpu_baseline and world == 1:               # rank 0 at N = 1 only
            n_s = min(args.cpu_sample, args.reads)
            if pe and n_s < min(args.batch, args.reads):
                # paired-end: mem_pestat is per batch, so a part of a batch is not the same problem (other insert-size bounds, other rescues)
                log(f"cpu sample raised from {n_s} to one whole batch ({min(args.batch, args.reads)} reads): paired-end parity needs the batch's own insert-size estimate")
                n_s = min(args.batch, args.reads)
            n_s = max(args.batch, n_s // args.batch * args.batch) if n_s >= args.batch else n_s & (~1 if pe else ~0)   # whole batches: the same -K cuts on both sides
            out["cpu_baseline"], sam_cpu_path = cpu_baseline(tp, prefix, reads[:n_s], pe, cpus, workdir, K_bases)
            if sam_cpu_path:
                # the CPU SAM and the GPU SAM of the same reads in the same run must be byte-identical.  The GPU side is the
                # file the stream driver wrote (its first n_s reads = whole batches with the same n_processed); without it, the resident pipeline's batch 0.
                if stream and os.path.exists(stream_sam_path):
                    parity_ok = file_prefix_equals(stream_sam_path, sam_cpu_path)
                    out["parity_checked"] = (f"{n_s} reads: the SAM file bwahip_stream_run wrote (FASTQ files -> SAM file, first {n_s} reads of {args.reads}) vs the SAM of "
                                             f"oracle/_ref/bwaref (the reference's own mem_process_seqs) on the same FASTQ reads with the same -K, byte for byte")
                    out["resident_sam_equals_stream_sam"] = file_prefix_equals(stream_sam_path, None, blob=sam_resident0)
                elif n_s == min(args.batch, args.reads):
                    parity_ok = file_equals(sam_cpu_path, sam_resident0)
                    out["parity_checked"] = f"{n_s} reads: SAM of the resident pipeline's batch 0 vs the SAM of the CPU path, byte for byte"
                out["parity_in_run"] = parity_ok
                if parity_ok is False:
                    log("PARITY FAILURE: GPU SAM differs from the reference CPU path's SAM on the bench workload")
        if other:
            out["other_configs"] = other
            if any(v.get("parity_in_run") is False for v in other.values()):
                parity_ok = False
        try:
            os.unlink(stream_sam_path)
        except OSError:
            pass
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if parity_ok is False:
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def other_kernels(km, counters, n_batches):
    sa_bytes = (64 * counters["lf"] + 8 * counters["sa"]) / n_batches
    return {
        "k_seeds": {"algorithmic_GBps": round(sa_bytes / (km["k_seeds"] * 1e-3) / 1e9, 1), "frac_of_peak": round(sa_bytes / (km["k_seeds"] * 1e-3) / 8e12, 4),
                    "what": "64 B per LF step + 8 B per SA read (bwt_sa, bwt.c:86), counted by the kernel; with the SA table kept for every row "
                            "(index_in_hbm.sa_intv = 1) there are no LF steps left and the stage is three short launches (rows, look-up, contig ids)"},
        "k_extend": {"GCUPS": round(counters["cells"] / n_batches / ((km["k_extend"] + km["k_extend_spec"]) * 1e-3) / 1e9, 2),
                     "what": "ksw_extend2 cell updates per second (cells counted by the kernel, k_extend + k_extend_spec time)"},
    }


def other_config(bw, tp, torch, dev, ctx, genome, lens, prefix, workdir, cpus, rl, sub_ppm, seed, genome_mbp, n_reads=1000000, n_parity=200000, steps=3):
    """A further BASELINE configuration on the index already in HBM: n_reads single-end reads as one batch, resident; single-context rate and
    per-kernel durations over `steps` passes, then the first n_parity reads through bwahip_process_seqs (host arrays in, SAM out) against the
    reference CPU path on the same reads."""
    t0 = time.time()
    reads = tp.make_reads(genome, lens, n_reads, rl, sub_ppm=sub_ppm, seed=seed, paired=False)
    res = Resident(bw, tp, torch, dev, reads, False, n_reads)
    opt = bw.default_opt()
    opt.n_threads = cpus
    res.step([ctx], opt)                                       # warm-up: buffers grow to this shape
    torch.cuda.synchronize()
    t1 = time.time()
    kms = []
    for _ in range(steps):
        res.step([ctx], opt, kms)
    torch.cuda.synchronize()
    el = time.time() - t1
    counters, _ = res.count(ctx, opt)
    km = {k: round(float(np.mean([x[k] for x in kms])), 3) for k in kms[0]}
    rec = {"workload": f"{n_reads} SE reads of {rl} bp, {sub_ppm / 10000:g}% substitutions, one batch, vs the same {genome_mbp} Mbp synthetic genome; {config_label(False, rl, sub_ppm)}",
           "single_context": {"value": round(n_reads * steps / el, 1), "unit": "reads/s", "ms_per_batch": round(el / steps * 1e3, 3), "steps": steps,
                              "what": "reads resident in HBM, SAM left in HBM, one context"},
           "kernel_ms": km, "per_read": per_read(counters, n_reads), "tail_us": tails(counters), "other_kernels": other_kernels(km, counters, 1)}
    k1 = km["k_smem"]
    alg = 64 * (counters["blocks"] - counters["heavy_blocks"] - counters["pass3_blocks"]) + n_reads * rl + 32 * (counters["intv"] - counters["heavy_intv"] - counters["pass3_intv"])
    rec["roofline"] = {"kernel": "k_smem", "bound": "hbm", "achieved": round(alg / (k1 * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (k1 * 1e-3) / 8e12, 4),
                       "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": k1}
    # parity + the CPU path's rate on the first n_parity reads
    names = tp.fixed_names(n_parity, False)
    t_e2e, sam_gpu = tp.process_seqs_bulk(bw, ctx, opt, names, reads[:n_parity])
    base, sam_cpu_path = cpu_baseline(tp, prefix, reads[:n_parity], False, cpus, workdir, 150000000, hot_path=False, tag=f"other_{rl}")
    rec["cpu_baseline"] = base
    rec["parity_in_run"] = file_equals(sam_cpu_path, sam_gpu) if sam_cpu_path else None
    rec["parity_checked"] = f"{n_parity} reads: SAM of bwahip_process_seqs vs SAM of the reference CPU path, byte for byte"
    if sam_cpu_path:
        os.unlink(sam_cpu_path)
    log(f"{config_label(False, rl, sub_ppm)}: {rec['single_context']['value']:.0f} reads/s single context, parity {rec['parity_in_run']} ({time.time() - t0:.1f}s)")
    del res
    torch.cuda.empty_cache()
    return rec


def kernel_src_sha():
    h = hashlib.sha256()
    for f in ("k_smem.hip", "fmi_dev.h"):
        h.update(open(os.path.join(ROOT, "bwa-mem-gpu_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def file_equals(path, blob):
    if os.path.getsize(path) != len(blob):
        return False
    return file_prefix_equals(path, None, blob=blob)


def file_prefix_equals(path, other_path, blob=None):
    """The first bytes of `path` equal the whole of `other_path` (or of blob)."""
    want = os.path.getsize(other_path) if other_path else len(blob)
    if os.path.getsize(path) < want:
        return False
    with open(path, "rb") as f, (open(other_path, "rb") if other_path else open(os.devnull, "rb")) as g:
        pos = 0
        while pos < want:
            n = min(1 << 24, want - pos)
            a = f.read(n)
            b = g.read(n) if other_path else blob[pos:pos + n]
            if a != b:
                return False
            pos += n
    return True


def measured_ceilings():
    """scripts/gather_bw (built by __graft_entry__.build()) in quick mode: ~1 s on the GPU, own process."""
    exe = os.path.join(ROOT, "scripts", "gather_bw")
    if not os.access(exe, os.X_OK):
        return None
    try:
        r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # the microbenchmark is an annotation, never a reason to lose the bench line
        log(f"gather_bw failed: {e}")
        return None


def cpu_baseline(tp, prefix, sample, pe, cpus, workdir, K_bases, hot_path=True, tag="cpu_sample"):
    """Time the CPU path on this node's cores on a bounded sample of the same workload: the reference's own sources
    (oracle/_ref/bwaref, kind "reference") when the prebuilt binary is present, else our C restatement (oracle/bwa_oracle,
    kind "port").  mem_process_seqs (align + finalisation + SAM text; its SAM is kept for the parity check), and with hot_path the same reads
    through the hot path alone (kt_for(worker1) == mem_align1_core per read, `-Z`).  Reported beside the GPU number; not a target."""
    n = len(sample)
    fqs = tp.write_fastq_fixed(os.path.join(workdir, tag), sample, pe)
    ref = os.path.join(ROOT, "oracle", "_ref", "bwaref")
    port = os.path.join(ROOT, "oracle", "bwa_oracle")
    exe, kind = (ref, "reference") if os.access(ref, os.X_OK) else (port, "port")
    sam_path = os.path.join(workdir, tag + ".sam")
    res = {"value": None, "unit": "reads/s", "cores": cpus, "kind": kind}

    def run(extra, stdout):
        # -K fixes the batch; without it the reference's own chunk_size * n_threads
        # (fastmap.c:304) can overflow int and degenerate to 1-read batches
        r = subprocess.run([exe, "mem", "-t", str(cpus), "-K", str(K_bases), *extra, prefix, *fqs], stdout=stdout, stderr=subprocess.PIPE, text=False)
        m = re.search(rb"aligned (\d+) reads in ([0-9.]+) s", r.stderr)
        return float(m.group(2)) if m and r.returncode == 0 else None, r.stderr[-300:]

    log(f"cpu baseline: {os.path.basename(exe)} on {n} reads with {cpus} threads (mem_process_seqs, SAM kept for the parity check) ...")
    with open(sam_path, "wb") as f:
        secs, err = run([], f)
    for fq in fqs:
        os.unlink(fq)
    if secs is None:
        res["sample"] = f"failed: {err!r}"
        return res, None
    res["value"] = round(n / secs, 1)
    res["seconds"] = round(secs, 2)
    res["sample"] = (f"{n} reads of the same workload ({'PE' if pe else 'SE'}) in batches of -K {K_bases} through mem_process_seqs (align + finalisation + SAM text) "
                     f"in {secs:.2f}s on {cpus} threads; index load excluded")
    if kind == "reference" and hot_path:
        fqs = tp.write_fastq_fixed(os.path.join(workdir, tag), sample, pe)
        log("cpu baseline: hot path only (kt_for(worker1)) ...")
        secs2, _ = run(["-Z"], subprocess.DEVNULL)
        for fq in fqs:
            os.unlink(fq)
        if secs2:
            res["hot_path_value"] = round(n / secs2, 1)
            res["hot_path_seconds"] = round(secs2, 2)
            res["hot_path_sample"] = (f"the same reads through kt_for(worker1) only (mem_align1_core per read, bwamem.c:1232) in {secs2:.2f}s, one run each: the difference to "
                                      f"{secs:.2f}s is within run-to-run noise when worker2 is a small share")
    return res, sam_path


if __name__ == "__main__":
    main()
