#!/bin/bash
# A/B of k_extend builds: per-kernel durations of the quick bench for each library variant
cd $GRAFT_REPO_ROOT
for v in "" _w6 _w5 _nw _nw5; do
  export BWAHIP_LIB=$GRAFT_REPO_ROOT/bwa-mem-gpu_amd/libbwahip$v.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ab$v; mkdir -p $OUT
  (cd /tmp; TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --genome-mbp 128 --reads 2000000 --steps 2 --warmup 1 --overlap 1 --no-cpu-baseline --no-e2e > $OUT/bench.json 2> $OUT/bench.log) || echo "failed $v"
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  echo "variant '$v': $(grep -E 'k_extend<3>|k_extend_spec<3>|k_dedup' $f | awk -F, '{printf "%s %.3f ms; ", substr($1,1,50), $4/1e6}')"
  python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print('   single', d['single_context']['value'], 'k_extend stage', d['kernel_ms']['k_extend'], 'ext_max', d['tail_us']['ext_max'])"
done
