#!/bin/bash
# per-kernel durations of the human-like bench point under rocprofv3: gpurun -- scripts/human_trace.sh <tag>
TAG=${1:-ht}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --genome-profile human-like --batch 1000000 --reads 1000000 --steps 1 --warmup 0 --overlap 1 --no-cpu-baseline --no-e2e --no-other-configs "$@" > $OUT/bench.json 2> $OUT/bench.log || echo "trace failed"
f=$(find $OUT/trace -name "*kernel_stats.csv" | xargs grep -l k_smem | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.1f} ms")
PY
cp "$f" $OUT/kernel_stats.csv
