#!/bin/bash
# exploratory: whole-path throughput with the batch split into sub-batches on concurrent streams (bench.py --lanes)
cd $GRAFT_REPO_ROOT
for L in 1 2 3 4; do
  timeout -k 10 300 python bench.py --lanes $L --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/lanes_$L.json 2> gpurun_out/lanes_$L.log || echo "lanes $L failed"
  python3 -c "
import json; d=json.load(open('gpurun_out/lanes_$L.json')); print('lanes', $L, d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['kernel_ms'].items() if v>1})"
done
