#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" _cw5 _cw8; do
  export BWAHIP_LIB=$GRAFT_REPO_ROOT/bwa-mem-gpu_amd/libbwahip$v.so
  timeout -k 10 400 python bench.py --genome-mbp 128 --reads 2000000 --steps 2 --warmup 1 --no-e2e > gpurun_out/abc$v.json 2> gpurun_out/abc$v.log || echo "failed $v"
  python3 -c "
import json; d=json.load(open('gpurun_out/abc$v.json')); print('variant $v: C3 k_cigar', d['kernel_ms']['k_cigar'], 'single', d['single_context']['value'], 'parity', d.get('parity_in_run'))
for k,v in d.get('other_configs',{}).items(): print('   ', k, v['single_context']['value'], v['parity_in_run'], 'k_cigar', v['kernel_ms']['k_cigar'])"
done
