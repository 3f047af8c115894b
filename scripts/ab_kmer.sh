#!/bin/bash
# interval table length sweep: gpurun -- scripts/ab_kmer.sh <tag> K...
TAG=${1:-abk}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
for K in "$@"; do
  BWAHIP_KMER_K=$K BWAHIP_VERBOSE=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > $OUT/k$K.json 2> $OUT/k$K.log || { echo "K=$K failed"; tail -5 $OUT/k$K.log; exit 1; }
  python3 - $OUT/k$K.json $K <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernel_ms"]
print("K", sys.argv[2], "single", d["single_context"]["value"], "pipeline", d.get("gpu_pipeline", {}).get("value"), "k_smem", k["k_smem"], "k_smem3", k["k_smem3"], "heavy", k["k_smem_heavy"], "frac", d["roofline"]["frac"], flush=True)
PY
done
grep -h "index in HBM" $OUT/k*.log | sort -u
