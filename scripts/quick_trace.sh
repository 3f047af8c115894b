#!/bin/bash
# per-kernel durations of the quick (128 Mbp) bench under rocprofv3: gpurun -- scripts/quick_trace.sh <tag> [bench args]
TAG=${1:-qt}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --genome-mbp 128 --reads 2000000 --steps 2 --warmup 1 --overlap 1 --no-cpu-baseline --no-e2e "$@" > $OUT/bench.json 2> $OUT/bench.log || echo "trace failed"
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.1f} ms")
PY
cp "$f" $OUT/kernel_stats.csv
