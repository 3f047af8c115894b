#!/usr/bin/env python3
"""Copy the summaries of a scripts/profile_round.sh run (gpurun_out/<tag>/) into profiles/<round>/ (tracked).
usage: scripts/collect_profiles.py <tag> <round-dir>"""
import csv, glob, hashlib, json, os, re, shutil, sys, collections

tag, dst = sys.argv[1], sys.argv[2]
traffic_only = "--traffic-only" in sys.argv       # on the GPU box: only (re)write profiles/pmc_traffic.json
src = os.path.join("gpurun_out", tag)
os.makedirs(dst, exist_ok=True)
if not traffic_only:
    if os.path.exists(os.path.join(src, "bench_default.json")):      # the unprofiled run may come from a later call (profile_round.sh bench)
        shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, "bench_default.json"))
    shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, "bench_under_rocprof_trace.json"))
    # rocprofv3 writes one set of files per process; bench.py also runs the gather microbenchmark as a child: take bench.py's
    ks = [f for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True) if "k_smem" in open(f).read()]
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "kernel_stats_bench_default.csv"))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(?:<\d+>)?)", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[m.group(1)][r["Counter_Name"]] += 1
if not traffic_only:
    with open(os.path.join(dst, "pmc_bench_default.csv"), "w") as fp:
        fp.write("kernel,counter,dispatches,sum_over_dispatches,per_dispatch\n")
        for k in sorted(tot):
            for c in sorted(tot[k]):
                fp.write(f"{k},{c},{calls[k][c]},{tot[k][c]:.6g},{tot[k][c] / calls[k][c]:.6g}\n")
# HBM traffic of the BWT-search kernel per launch (FETCH_SIZE / WRITE_SIZE are in KiB; calibration in profiles README)
key = [k for k in tot if k.startswith("k_smem") and "heavy" not in k]
if key:
    t = tot[key[0]]; n = calls[key[0]]
    bench = json.load(open(os.path.join(src, "bench_under_trace.json")))
    sha = hashlib.sha256()
    for f in ("k_smem.hip", "fmi_dev.h"):
        sha.update(open(os.path.join("bwa-mem-gpu_amd", "csrc", f), "rb").read())
    out = {"launch_workload": bench["config"]["launch_workload"], "kernel": key[0], "kernel_src_sha256": sha.hexdigest(),
           "fetch_bytes_per_launch": t["FETCH_SIZE"] / n["FETCH_SIZE"] * 1024, "write_bytes_per_launch": t["WRITE_SIZE"] / n["WRITE_SIZE"] * 1024,
           "source": f"{dst}/pmc_bench_default.csv (separate rocprofv3 --pmc passes of `python3 bench.py --steps 1 --warmup 0 --reads 2000000 --overlap 1 --no-cpu-baseline --no-e2e`; per launch = per batch of 1 M reads)"}
    json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
    print(out)
