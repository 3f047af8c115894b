#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X BWA-MEM hot path on BASELINE.json's headline configuration: 150 bp paired-end
reads (2x150, insert N(500,50^2), FR) against an hg38-scale genome (3.1 Gbp; GRCh38 itself is not available offline, so a
deterministic synthetic genome with repeat families is used and named as such), with the roofline of the BWT-search
kernel, the end-to-end rate (host buffers in -> SAM text out), a byte-for-byte check of that SAM against the reference's
own CPU path on the same reads, and the CPU path timed on this node's cores in the same run.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the whole GPU pipeline -- mem_process_seqs end to end: hot path (mem_align1_core), insert-size statistics,
mate rescue, pairing, mark-primary, CIGAR/NM/MD/mapQ, SAM text -- over this rank's resident read set (default 10 M reads = 5 M
pairs, BASELINE configs[2]), processed in batches of 1 M reads exactly as `bwa mem -K 150000000` would cut them; inputs (base
codes, names, qualities) are in HBM when the timed region starts, the SAM text stays in HBM.  Reads are sharded per rank with no data-path collective
("weak" scaling: every rank aligns its own reads); RCCL is used once, to broadcast the index from rank 0.
K steps are timed twice, each time bracketed by barrier + synchronize: one batch at a time on one context (`single_context`; the
per-kernel durations and the roofline come from this region), then double buffered on --overlap contexts (`value`).
Rank 0 prints ONE JSON line.
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
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BWAHIP_BENCH_READS", "10000000")),
                    help="reads per GPU and step (PE: mates interleaved, so half as many pairs)")
    ap.add_argument("--batch", type=int, default=1000000, help="reads per mem_process_seqs batch (bwa mem -K 150000000 at 150 bp)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--se", action="store_true", help="single-end reads (BASELINE configs[1]) instead of paired-end")
    ap.add_argument("--sub-ppm", type=int, default=10000, help="substitution errors per million bases (configs[4]: 50000)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BWAHIP_BENCH_CPU_READS", "1000000")),
                    help="reads of the same workload given to the CPU path (and compared byte for byte with the GPU path's SAM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("BWAHIP_BENCH_OVERLAP", "2")),
                    help="contexts per GPU taking the batches of a step in turn (double buffering) for `value`; 1 = one batch at a time only")
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
    prefix = os.path.join(workdir, f"g{args.genome_mbp}")
    fa = prefix + ".fa"
    t_index = 0.0
    lens = tp.contig_lengths(args.genome_mbp * 1000000)
    cpus = usable_cpus()
    # ---------------- genome + index: rank 0 builds (stock bwa format, cached under /dev/shm), the others receive it over RCCL
    genome = None
    if rank == 0:
        t0 = time.time()
        if not (os.path.exists(prefix + ".sa") and os.path.exists(fa)):
            genome = tp.make_genome(38, lens, repeats=True)
            tp.write_fasta(fa, genome, lens)
            log(f"genome {args.genome_mbp} Mbp generated and written: {time.time() - t0:.1f}s")
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
            index_distribution = "bwahip_init_rccl (ncclBroadcast x4 inside libbwahip.so)"
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

    # ---------------- reads: every rank its own set (seed 103 + rank), 1 % substitutions, 50 % reverse strand; PE: FR pairs,
    # insert N(500, 50^2) clipped to [300, 700], mates interleaved.  The genome bytes come from the FASTA rank 0 wrote.
    if genome is None:
        genome = tp.read_fasta_bases(fa, lens)
    t0 = time.time()
    reads = tp.make_reads(genome, lens, args.reads, args.read_len, sub_ppm=args.sub_ppm, seed=(103 if pe else 102) + rank, paired=pe)
    log(f"rank {rank}: {args.reads} reads generated ({time.time() - t0:.1f}s)")
    rl = args.read_len
    n_batches = (args.reads + args.batch - 1) // args.batch
    codes_dev = torch.empty(args.reads * rl, dtype=torch.uint8, device=dev)
    for b0 in range(0, args.reads, 2000000):                   # staged: the ASCII -> code table look-up doubles the host footprint otherwise
        b1 = min(args.reads, b0 + 2000000)
        codes_dev[b0 * rl:b1 * rl] = torch.from_numpy(bw.NT4[reads[b0:b1].reshape(-1)]).to(dev)
    off_dev = torch.arange(args.batch + 1, dtype=torch.int64, device=dev) * rl
    # the text the SAM stage prints, resident too: names (fixed width, NUL terminated), qualities (all 'I': one batch worth, shared)
    names_host = tp.fixed_names(args.reads, pe)
    nw = names_host.shape[1]
    names_dev = torch.zeros(args.reads * nw + 64, dtype=torch.uint8, device=dev)
    names_dev[:args.reads * nw] = torch.from_numpy(names_host.reshape(-1)).to(dev)
    name_off_dev = torch.arange(args.batch + 1, dtype=torch.int64, device=dev) * nw
    qual_dev = torch.full((args.batch * rl + 64,), ord("I"), dtype=torch.uint8, device=dev)
    qual_off_dev = torch.arange(args.batch, dtype=torch.int64, device=dev) * rl
    torch.cuda.synchronize()
    opt = bw.default_opt()
    if pe:
        opt.flag |= 0x2
    opt.n_threads = max(1, cpus // world)

    def batch_bounds(b):
        b0 = b * args.batch
        return b0, min(args.reads, b0 + args.batch)

    def attach(b0, b1, cx=None):
        cx = cx or ctx
        cx.batch_attach(b1 - b0, codes_dev.data_ptr() + b0 * rl, off_dev.data_ptr(), rl, (b1 - b0) * rl)
        cx.batch_attach_text(qual_dev.data_ptr(), qual_off_dev.data_ptr(), names_dev.data_ptr() + b0 * nw, name_off_dev.data_ptr())

    def run_step(collect=None, ctxs=None):
        """One pass over this rank's reads, batch by batch.  With several contexts (double buffering) context t takes batches t, t + n, ...
        on its own streams, driven by its own host thread."""
        ctxs = ctxs or [ctx]

        def work(t):
            for b in range(t, n_batches, len(ctxs)):
                b0, b1 = batch_bounds(b)
                attach(b0, b1, ctxs[t])
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

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        kms = []
        run_step(kms)
        log(f"warmup {w}: kernel ms of batch 0 {dict((k, round(v, 2)) for k, v in kms[0].items())}")
    sync_all()
    t0 = time.time()
    kms = []
    for _ in range(args.steps):
        run_step(kms)
    sync_all()
    elapsed = time.time() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # ---------------- the same K steps again with double buffering: --overlap contexts (sharing the index in HBM) take the batches in turn.
    # This is the production schedule and the headline `value`; the one-batch-at-a-time region above stays in the line as
    # `single_context`, and it is the one the per-kernel durations and the roofline are measured in (a kernel that shares the GPU
    # with another batch's kernels has no duration of its own).
    dbuf = None
    if args.overlap > 1 and n_batches > 1:
        ctxs = [ctx] + [ctx.clone() for _ in range(args.overlap - 1)]
        for _ in range(max(1, min(args.warmup, 2))):
            run_step(None, ctxs)
        sync_all()
        t0 = time.time()
        for _ in range(args.steps):
            run_step(None, ctxs)
        sync_all()
        el2 = time.time() - t0
        if world > 1:
            tmax = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el2 = float(tmax.item())
        dbuf = {"value": round(args.reads * world * args.steps / el2, 1), "unit": "reads/s", "ms_per_step": round(el2 / args.steps * 1e3, 3), "contexts_per_gpu": args.overlap,
                "what": "`value`: the batches of a step are taken in turn by several contexts on the GPU (bwahip_ctx_clone: shared index, own streams and "
                        "host thread each), so that one batch's latency-bound kernels and serial tails run under another batch's throughput-bound kernels"}
        for c2 in ctxs[1:]:
            c2.close()
        log(f"double buffered ({args.overlap} contexts): {dbuf['value']:.0f} reads/s vs {args.reads * world * args.steps / elapsed:.0f} one batch at a time")
    # algorithmic work per launch, counted by the kernels themselves: one more (untimed) pass, counters read after every batch
    cnts = []
    sam_bytes = 0
    sam_resident0 = None
    for b in range(n_batches):
        b0, b1 = batch_bounds(b)
        attach(b0, b1)
        ctx.batch_run_sam(opt, n_processed=b0)
        cnts.append(ctx.counters())
        if b == 0:
            sam_resident0 = ctx.batch_sam()
            sam_bytes = len(sam_resident0)
            if pe:
                pes_b0 = ctx.last_pe_stats()[0]
                log(f"batch 0: insert size FR {pes_b0[1]}; mate rescue {ctx.pe_counters}")
    counters = {k: (max(c[k] for c in cnts) if k.endswith("_max") or k.startswith("max_") else sum(c[k] for c in cnts)) for k in cnts[0]}

    # ---------------- end to end (host buffers in -> SAM text out) on the first --cpu-sample reads, one mem_process_seqs batch
    n_s = min(args.cpu_sample, args.reads) & (~1 if pe else ~0)
    e2e = None
    sam_gpu = None
    if not args.no_e2e:
        t_e2e, sam_gpu = tp.process_seqs_bulk(bw, ctx, opt, names_host[:n_s], reads[:n_s])     # first call: buffers grow to the batch size
        t_rest = []
        for _ in range(3):                                         # steady state, as in a run over many batches
            t_again, sam_again = tp.process_seqs_bulk(bw, ctx, opt, names_host[:n_s], reads[:n_s])
            assert sam_again == sam_gpu
            t_rest.append(t_again)
        del sam_again
        log(f"e2e: first call {t_e2e:.3f}s, then {' '.join(f'{t:.3f}' for t in t_rest)}s")
        t_e2e = min([t_e2e] + t_rest)
        if world > 1:
            tm = torch.tensor([t_e2e], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            t_e2e = float(tm.item())
        e2e = {"reads_per_s": n_s * world / t_e2e, "seconds": t_e2e, "reads": n_s * world, "host_threads_per_gpu": opt.n_threads}
        log(f"e2e: {n_s} reads through bwahip_process_seqs in {t_e2e:.3f}s with {opt.n_threads} host threads")
        if world == 1:
            # the same batch with the SAM handed over in one piece (bwahip_process_seqs_text: what a caller whose output step is one fwrite
            # uses), and two such batches in flight on two contexts (bwahip_ctx_clone), each caller with half of the host threads
            import zlib
            want = (len(sam_gpu), zlib.crc32(sam_gpu))
            one = tp.bulk_caller(bw, ctx, opt, names_host[:n_s], reads[:n_s], one_piece=True)
            _, ln, crc = one(check=True)
            assert (ln, crc) == want, "bwahip_process_seqs_text: SAM differs from bwahip_process_seqs'"
            e2e["one_piece_reads_per_s"] = n_s / min(one()[0] for _ in range(3))
            log(f"e2e, SAM in one piece: {e2e['one_piece_reads_per_s']:.0f} reads/s")
            if args.overlap > 1:
                c2 = ctx.clone()
                o2 = bw.Opt.from_buffer_copy(opt)
                o2.n_threads = max(1, opt.n_threads // 2)
                callers = [tp.bulk_caller(bw, cx, o2, names_host[:n_s], reads[:n_s], one_piece=True) for cx in (ctx, c2)]
                for c in callers:
                    _, ln, crc = c(check=True)
                    assert (ln, crc) == want, "SAM of a batch differs on the second context"
                lens = [[], []]

                def caller(t):
                    for _ in range(4):
                        lens[t].append(callers[t]()[1])
                th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
                t0 = time.time()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                t2 = time.time() - t0
                c2.close()
                assert all(v == want[0] for v in lens[0] + lens[1])
                e2e["two_in_flight_reads_per_s"] = 8 * n_s / t2
                log(f"e2e, two batches in flight: {8 * n_s} reads in {t2:.3f}s")

    if rank == 0:
        n_launch = len(kms)
        ms_step = elapsed / args.steps * 1e3
        value = args.reads * world * args.steps / elapsed
        single = {"value": round(value, 1), "unit": "reads/s", "ms_per_step": round(ms_step, 3), "steps": args.steps,
                  "what": "the same steps, one batch at a time on one context: the timed region `kernel_ms` and `roofline` are measured in"}
        if dbuf:
            value, ms_step = dbuf["value"], dbuf["ms_per_step"]
        k1 = float(np.mean([k["k_smem"] for k in kms]))
        # algorithmic bytes of the BWT-search kernel per launch: 64 B per Occ block touched by bwt_extend +
        # the read bytes in + 32 B per interval out, counted by the kernel itself (the few reads k_smem hands to
        # k_smem_heavy, and pass 3 which runs in k_smem3, are counted by those kernels and subtracted)
        alg_total = (64 * (counters["blocks"] - counters["heavy_blocks"] - counters["pass3_blocks"]) + args.reads * rl +
                     32 * (counters["intv"] - counters["heavy_intv"] - counters["pass3_intv"]))
        alg_bytes = alg_total / n_batches
        achieved = alg_bytes / (k1 * 1e-3) / 1e9
        kind = (f"{args.reads // 2} pairs (2x{rl} bp, FR, insert N(500,50^2) clipped [300,700])" if pe else f"{args.reads} SE reads of {rl} bp")
        out = {
            "metric": f"reads/s aligned ({rl} bp {'PE' if pe else 'SE'} vs hg38-scale {args.genome_mbp} Mbp synthetic genome), GPU pipeline with reads resident in HBM",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{kind}, {args.sub_ppm / 10000:g}% substitutions, per GPU and step, in batches of {args.batch} reads (-K {args.batch * rl}) vs "
                                   f"{args.genome_mbp} Mbp synthetic genome with repeat families (GRCh38 not available offline); "
                                   f"BASELINE configs[{2 if pe else 1}] shape",
                       "launch_workload": f"one batch of {args.batch} {'PE' if pe else 'SE'} reads of {rl} bp, {args.sub_ppm / 10000:g}% substitutions vs {args.genome_mbp} Mbp synthetic genome",
                       "reads_per_gpu": args.reads, "batch_reads": args.batch, "read_len": rl, "paired": pe, "genome_mbp": args.genome_mbp,
                       "stages": ctx.stage_names(),
                       "output": ctx.output_description(pe),
                       "index_build_s": round(t_index, 1), "index_to_hbm_s": round(t_bcast, 2), "index_distribution": index_distribution, "host_cpus_usable": cpus},
            "kernel_ms": {k: round(float(np.mean([x[k] for x in kms])), 3) for k in kms[0]},
            "launches_timed": n_launch, "sam_bytes_per_batch": sam_bytes,
            "per_read": {"bwt_extend": round(counters["extend"] / args.reads, 1), "occ_blocks": round(counters["blocks"] / args.reads, 1),
                         "sa_lookups": round(counters["sa"] / args.reads, 2), "lf_steps": round(counters["lf"] / args.reads, 1),
                         "dp_cells": round(counters["cells"] / args.reads, 1),
                         "dp_rows_1col": round(counters["dp_rows_1col"] / args.reads, 1), "dp_rows_ncol": round(counters["dp_rows_ncol"] / args.reads, 1)},
            "tail_us": {k: round(v / 100.0, 1) for k, v in counters.items() if k.endswith("_max")},
            "tail_counts": {"max_extends_per_read": counters.get("max_extends"), "max_seeds_per_read": counters.get("max_seeds"), "max_chains_per_read": counters.get("max_chains")},
            "roofline": {"kernel": "k_smem", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k1, 3), "reads_per_launch": args.batch},
        }
        out["single_context"] = single
        # the other kernels SURVEY.md 8(d) asks figures for: the SA look-up kernels against the same HBM roofline, the integer DP of the
        # extension in cell updates per second (no roofline fraction: VALU / LDS issue bound, no MFMA work anywhere)
        km = out["kernel_ms"]
        sa_bytes = (64 * counters["lf"] + 8 * counters["sa"]) / n_batches
        out["other_kernels"] = {
            "k_seeds": {"algorithmic_GBps": round(sa_bytes / (km["k_seeds"] * 1e-3) / 1e9, 1), "frac_of_peak": round(sa_bytes / (km["k_seeds"] * 1e-3) / 8e12, 4),
                        "what": "64 B per LF step + 8 B per SA read (bwt_sa, bwt.c:86), counted by the kernel"},
            "k_extend": {"GCUPS": round(counters["cells"] / n_batches / ((km["k_extend"] + km["k_extend_spec"]) * 1e-3) / 1e9, 2),
                         "what": "ksw_extend2 cell updates per second (cells counted by the kernel, k_extend + k_extend_spec time)"},
        }
        if dbuf:
            out["schedule"] = dbuf
            out["sum_kernel_ms_over_ms_per_batch"] = round(sum(out["kernel_ms"].values()) / (ms_step / n_batches), 3)
        if e2e:
            out["value_e2e"] = round(e2e["reads_per_s"], 1)
            out["e2e"] = {"what": "bwahip_process_seqs: host bseq1_t arrays in (ASCII reads, names, qualities) -> seqs[i].sam text out, one batch per GPU, PCIe and host work included",
                          "reads": e2e["reads"], "seconds": round(e2e["seconds"], 4), "host_threads_per_gpu": e2e["host_threads_per_gpu"]}
            if "one_piece_reads_per_s" in e2e:
                out["e2e"]["one_piece_reads_per_s"] = round(e2e["one_piece_reads_per_s"], 1)
                out["e2e"]["one_piece_what"] = "bwahip_process_seqs_text: the same call with the batch's SAM handed over as one buffer (no malloc per read); same bytes"
            if "two_in_flight_reads_per_s" in e2e:
                out["e2e"]["two_in_flight_reads_per_s"] = round(e2e["two_in_flight_reads_per_s"], 1)
                out["e2e"]["two_in_flight_what"] = ("two caller threads, each with its own context (bwahip_ctx_clone: shared index) and half of the host threads, "
                                                    "4 batches each back to back through bwahip_process_seqs_text")
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
        if sam_gpu is not None and sam_resident0 is not None and n_s == min(args.batch, args.reads):
            out["resident_sam_equals_e2e_sam"] = sam_resident0 == sam_gpu     # the timed pipeline's own output for batch 0 is what the parity check covers
        parity_ok = None
        if not args.no_cpu_baseline and world == 1:               # rank 0 at N = 1 only
            out["cpu_baseline"], sam_cpu_path = cpu_baseline(tp, prefix, reads[:n_s], pe, cpus, workdir)
            if sam_gpu is not None and sam_cpu_path:
                # the CPU SAM and the GPU SAM of the same reads in the same run must be byte-identical
                parity_ok = file_equals(sam_cpu_path, sam_gpu)
                out["parity_in_run"] = parity_ok
                out["parity_checked"] = f"{n_s} reads: SAM of bwahip_process_seqs vs SAM of oracle/_ref/bwaref (the reference's own mem_process_seqs), byte for byte"
                if not parity_ok:
                    log("PARITY FAILURE: GPU SAM differs from the reference CPU path's SAM on the bench workload")
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if parity_ok is False:
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_src_sha():
    h = hashlib.sha256()
    for f in ("k_smem.hip", "fmi_dev.h"):
        h.update(open(os.path.join(ROOT, "bwa-mem-gpu_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def file_equals(path, blob):
    if os.path.getsize(path) != len(blob):
        return False
    with open(path, "rb") as f:
        pos = 0
        while True:
            chunk = f.read(1 << 24)
            if not chunk:
                return True
            if chunk != blob[pos:pos + len(chunk)]:
                return False
            pos += len(chunk)


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


def cpu_baseline(tp, prefix, sample, pe, cpus, workdir):
    """Time the CPU path on this node's cores on a bounded sample of the same workload: the reference's own sources
    (oracle/_ref/bwaref, kind "reference") when the prebuilt binary is present, else our C restatement (oracle/bwa_oracle,
    kind "port").  Two runs: mem_process_seqs (align + finalisation + SAM text; its SAM is kept for the parity check) and
    the hot path alone (kt_for(worker1) == mem_align1_core per read, `-Z`), which is what the GPU `value` covers.
    Reported beside the GPU number; not a target."""
    n = len(sample)
    fqs = tp.write_fastq_fixed(os.path.join(workdir, "cpu_sample"), sample, pe)
    ref = os.path.join(ROOT, "oracle", "_ref", "bwaref")
    port = os.path.join(ROOT, "oracle", "bwa_oracle")
    exe, kind = (ref, "reference") if os.access(ref, os.X_OK) else (port, "port")
    sam_path = os.path.join(workdir, "cpu_sample.sam")
    res = {"value": None, "unit": "reads/s", "cores": cpus, "kind": kind}

    def run(extra, stdout):
        # -K fixes the batch at 150 M bases; without it the reference's own chunk_size * n_threads
        # (fastmap.c:304) can overflow int and degenerate to 1-read batches
        r = subprocess.run([exe, "mem", "-t", str(cpus), "-K", "150000000", *extra, prefix, *fqs], stdout=stdout, stderr=subprocess.PIPE, text=False)
        m = re.search(rb"aligned (\d+) reads in ([0-9.]+) s", r.stderr)
        return float(m.group(2)) if m and r.returncode == 0 else None, r.stderr[-300:]

    log(f"cpu baseline: {os.path.basename(exe)} on {n} reads with {cpus} threads (mem_process_seqs, SAM kept for the parity check) ...")
    with open(sam_path, "wb") as f:
        secs, err = run([], f)
    if secs is None:
        res["sample"] = f"failed: {err!r}"
        return res, None
    res["value"] = round(n / secs, 1)
    res["sample"] = (f"{n} reads of the same workload ({'PE' if pe else 'SE'}) as one batch through mem_process_seqs (align + finalisation + SAM text) "
                     f"in {secs:.2f}s on {cpus} threads; index load excluded")
    if kind == "reference":
        log("cpu baseline: hot path only (kt_for(worker1)) ...")
        secs2, _ = run(["-Z"], subprocess.DEVNULL)
        if secs2:
            res["hot_path_value"] = round(n / secs2, 1)
            res["hot_path_sample"] = f"the same reads through kt_for(worker1) only (mem_align1_core per read, bwamem.c:1232) in {secs2:.2f}s (the share of the CPU path that mem_align1_core is; `value` covers all of mem_process_seqs, like `cpu_baseline.value`)"
    return res, sam_path


if __name__ == "__main__":
    main()
