#!/bin/bash
# usage: scripts/env_sweep.sh VAR v1 v2 ...  -> short 128 Mbp bench per value
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --genome-mbp 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sweep_$v.json 2> gpurun_out/sweep_$v.log
  python3 -c "
import json; d=json.load(open('gpurun_out/sweep_$v.json')); print('$VAR=$v', d['value'], {k: round(x,2) for k,x in d['kernel_ms'].items() if x > 1}, d['tail_us']['chain_build_max'], d['tail_us']['chain_sort_max'])"
done
