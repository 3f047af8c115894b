#!/bin/bash
# A/B: current library against one with round 2's k_extend.hip (fused dedup, no window), full-size genome, one context, per-kernel durations
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --reads 2000000 > gpurun_out/ab_warm.json 2> gpurun_out/ab_warm.log   # builds the index
for v in "" _old ""  _old; do
  export BWAHIP_LIB=$GRAFT_REPO_ROOT/bwa-mem-gpu_amd/libbwahip$v.so
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-e2e --reads 3000000 --overlap 1 > gpurun_out/abo$v.json 2> gpurun_out/abo$v.log || echo "failed $v"
  python3 -c "
import json; d=json.load(open('gpurun_out/abo$v.json')); k=d['kernel_ms']; print('variant [$v]: single', d['single_context']['value'], 'k_extend', k['k_extend'], 'spec', k['k_extend_spec'], 'k_chain', k['k_chain'], 'sum', round(sum(k.values()),1), 'tails', d['tail_us']['ext_max'], d['tail_us']['ext_dedup_max'])"
done
