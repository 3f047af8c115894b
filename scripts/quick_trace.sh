#!/bin/bash
# kernel stats of a short bench run (2 M reads, no e2e / CPU legs):  gpurun -- scripts/quick_trace.sh <tag> [bench args]
TAG=${1:-qt}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --reads 2000000 --no-cpu-baseline --no-e2e "$@" > $OUT/bench.json 2> $OUT/trace.log || echo "trace failed"
f=$(grep -l k_smem $OUT/trace/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
cut -d, -f1-4 $OUT/kernel_stats.csv | head -40
