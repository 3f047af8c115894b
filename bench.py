#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X BWA-MEM hot path (mem_align1_core on the GPU: SMEM collection, SA
look-up, chaining + filter, banded-SW extension, dedup/patch), with the roofline of the BWT-search kernel
and a CPU baseline timed on this node's cores in the same run.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the whole hot path over one resident batch (codes in HBM -> alignment regions in
HBM).  Reads are sharded per rank with no data-path collective ("weak" scaling: every rank aligns its own
batch); RCCL is used once, to broadcast the index from rank 0.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=int, default=int(os.environ.get("BWAHIP_BENCH_MBP", "512")),
                    help="size of the synthetic genome (GRCh38 itself is not available offline)")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("BWAHIP_BENCH_READS", "1000000")))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub-ppm", type=int, default=10000, help="substitution errors per million bases (configs[4]: 50000)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BWAHIP_BENCH_CPU_READS", "200000")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=1,
                    help="exploratory: split the batch into this many sub-batches, each with its own context and stream, run "
                         "concurrently (stages of different sub-batches overlap; per-kernel timings then overlap too)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
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
    # ---------------- index: rank 0 builds (stock bwa format), the others receive it over RCCL
    if rank == 0:
        t0 = time.time()
        genome = tp.make_genome(38, lens, repeats=True)
        if not os.path.exists(prefix + ".sa"):
            tp.write_fasta(fa, genome, lens)
            bw.make_index(fa, prefix)
        t_index = time.time() - t0
        log(f"genome {args.genome_mbp} Mbp + index: {t_index:.1f}s")
    if world > 1:
        dist.barrier()
    t0 = time.time()
    if rank == 0 or world == 1:
        ctx = bw.Context(prefix, local_rank)
        holder = None
    if world > 1:
        ctx_or_holder = tp.broadcast_index(bw, dist, torch, prefix if rank == 0 else None, rank, local_rank)
        if rank != 0:
            ctx, holder = ctx_or_holder
    t_bcast = time.time() - t0

    # ---------------- reads: every rank its own batch (seed 102 + rank), 1 % substitutions, 50 % reverse strand
    if rank != 0:
        genome = tp.make_genome(38, lens, repeats=True)
    reads = tp.make_reads(genome, lens, args.reads, args.read_len, sub_ppm=args.sub_ppm, seed=102 + rank)
    codes = bw.NT4[reads.reshape(-1)]
    off = np.arange(args.reads + 1, dtype=np.int64) * args.read_len
    log(f"rank {rank}: {args.reads} reads generated")
    opt = bw.default_opt()
    lanes = None
    if args.lanes > 1:
        # sub-batches: contexts that adopt one shared copy of the index (bwahip_init_device), one host thread each
        from concurrent.futures import ThreadPoolExecutor
        meta, arrays = tp.load_index_arrays(prefix)
        shared = {k: torch.from_numpy(v).to(f"cuda:{local_rank}") for k, v in arrays.items()}
        torch.cuda.synchronize()
        lanes = [bw.Context.from_device_arrays(meta, shared["bwt"].data_ptr(), shared["sa"].data_ptr(), shared["pac"].data_ptr(), local_rank)
                 for _ in range(args.lanes)]
        per = (args.reads + args.lanes - 1) // args.lanes
        for i, lc in enumerate(lanes):
            b0, b1 = i * per, min(args.reads, (i + 1) * per)
            lc.batch_upload(codes[b0 * args.read_len:b1 * args.read_len], off[b0:b1 + 1] - off[b0])
        pool = ThreadPoolExecutor(args.lanes)

        class Lanes:                                    # same surface as one context for the loop below
            def batch_run(self, o):
                return list(pool.map(lambda lc: lc.batch_run(o), lanes))[0]

            def counters(self):
                tot = None
                for lc in lanes:
                    cs = lc.counters()
                    tot = cs if tot is None else {k: (max(tot[k], v) if k.endswith("_max") or k.startswith("max_") else tot[k] + v) for k, v in cs.items()}
                return tot
        ctx = Lanes()
    else:
        ctx.batch_upload(codes, off)
    log("batch uploaded")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        km = ctx.batch_run(opt)
        log(f"warmup {w}: kernel ms {km}")
    sync_all()
    t0 = time.time()
    kms = []
    for _ in range(args.steps):
        kms.append(ctx.batch_run(opt))
    sync_all()
    elapsed = time.time() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    counters = ctx.counters()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = args.reads * world * args.steps / elapsed
        k1 = float(np.mean([k["k_smem"] for k in kms]))
        # algorithmic bytes of the BWT-search kernel per launch: 64 B per Occ block touched by
        # bwt_extend + the read bytes in + 32 B per interval out, counted by the kernel itself
        # (the few reads k_smem hands to k_smem_heavy are counted by that kernel and subtracted here)
        # (pass 3 runs in k_smem3; its blocks and intervals are counted by that kernel and subtracted too)
        alg_bytes = (64 * (counters["blocks"] - counters["heavy_blocks"] - counters["pass3_blocks"]) + args.reads * args.read_len +
                     32 * (counters["intv"] - counters["heavy_intv"] - counters["pass3_intv"]))
        achieved = alg_bytes / (k1 * 1e-3) / 1e9
        out = {
            "metric": "reads/s aligned (150 bp vs hg38-scale synthetic genome), hot path mem_align1_core on GPU",
            "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.reads} synthetic {args.read_len} bp SE reads ({args.sub_ppm / 10000:g}% substitutions) per GPU vs "
                                   f"{args.genome_mbp} Mbp synthetic genome with repeat families (GRCh38 not available offline); "
                                   "BASELINE configs[1] shape",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "genome_mbp": args.genome_mbp, "lanes": args.lanes,
                       "stages": ["k_smem(passes 1-2)+k_smem_heavy+k_smem3(pass 3)+k_intv_sort", "k_seeds", "k_chain", "k_extend_spec+k_extend(+dedup/patch)"],
                       "output": "mem_alnreg_v per read resident in HBM (== mem_align1_core)",
                       "index_build_s": round(t_index, 1), "index_broadcast_s": round(t_bcast, 2)},
            "kernel_ms": {k: round(float(np.mean([x[k] for x in kms])), 3) for k in kms[0]},
            "per_read": {"bwt_extend": round(counters["extend"] / args.reads, 1), "occ_blocks": round(counters["blocks"] / args.reads, 1),
                         "sa_lookups": round(counters["sa"] / args.reads, 2), "lf_steps": round(counters["lf"] / args.reads, 1),
                         "dp_cells": round(counters["cells"] / args.reads, 1),
                         "dp_rows_1col": round(counters["dp_rows_1col"] / args.reads, 1), "dp_rows_ncol": round(counters["dp_rows_ncol"] / args.reads, 1)},
            "tail_us": {k: round(v / 100.0, 1) for k, v in counters.items() if k.endswith("_max")},
            "tail_counts": {"max_extends_per_read": counters.get("max_extends"), "max_seeds_per_read": counters.get("max_seeds"), "max_chains_per_read": counters.get("max_chains")},
            "roofline": {"kernel": "k_smem", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(k1, 3)},
        }
        # HBM traffic of the same kernel from the committed PMC pass of this very command (counters cannot be read from inside
        # the process); only quoted when the workload string matches
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == out["config"]["workload"]:
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
        if not args.no_cpu_baseline and world == 1:               # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(tp, bw, prefix, genome, lens, args, workdir)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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


def cpu_baseline(tp, bw, prefix, genome, lens, args, workdir):
    """Time the CPU path on this node's cores on a bounded sample of the same workload: the reference's own
    sources (oracle/_ref/bwaref, kind "reference") when the prebuilt binary is present, else our C restatement
    (oracle/bwa_oracle, kind "port").  Reported beside the GPU number; not a target."""
    n = min(args.cpu_sample, args.reads)
    cores = os.cpu_count() or 1
    fq = os.path.join(workdir, "cpu_sample.fq")
    reads = tp.make_reads(genome, lens, n, args.read_len, sub_ppm=args.sub_ppm, seed=102)
    tp.write_fastq(fq, reads)
    ref = os.path.join(ROOT, "oracle", "_ref", "bwaref")
    port = os.path.join(ROOT, "oracle", "bwa_oracle")
    exe, kind = (ref, "reference") if os.access(ref, os.X_OK) else (port, "port")
    # -K fixes the batch at 150 M bases; without it the reference's own
    # chunk_size * n_threads (fastmap.c:304) overflows int at this core count and degenerates to 1-read batches
    log(f"cpu baseline: {os.path.basename(exe)} on {n} reads with {cores} threads ...")
    r = subprocess.run([exe, "mem", "-t", str(cores), "-K", "150000000", prefix, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    m = re.search(r"aligned (\d+) reads in ([0-9.]+) s", r.stderr)
    if not m:
        return {"value": None, "unit": "reads/s", "cores": cores, "kind": kind, "sample": f"failed: {r.stderr[-200:]}"}
    secs = float(m.group(2))
    return {"value": round(n / secs, 1), "unit": "reads/s", "cores": cores, "kind": kind,
            "sample": f"{n} reads of the same workload through mem_process_seqs (align + SAM text) in {secs:.2f}s; index load excluded"}


if __name__ == "__main__":
    main()
