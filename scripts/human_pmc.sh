#!/bin/bash
# SQ counters of the human-like bench point, one pass: gpurun -- scripts/human_pmc.sh <tag>
TAG=${1:-hp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --genome-profile human-like --batch 1000000 --reads 1000000 --steps 1 --warmup 0 --overlap 1 --no-cpu-baseline --no-e2e --no-other-configs > $OUT/bench.json 2> $OUT/bench.log || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m: tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CYCLES", 0))[:14]:
    c = tot[k]
    print(f"{k:28s} VALU {c.get('SQ_INSTS_VALU',0):.3g} SALU {c.get('SQ_INSTS_SALU',0):.3g} LDS {c.get('SQ_INSTS_LDS',0):.3g} VMEM_RD {c.get('SQ_INSTS_VMEM_RD',0):.3g} waves {c.get('SQ_WAVES',0):.3g} wave_cycles {c.get('SQ_WAVE_CYCLES',0):.3g} wait_any {c.get('SQ_WAIT_ANY',0):.3g} busy {c.get('SQ_BUSY_CYCLES',0):.3g}")
PY
