#!/bin/bash
# file-to-file rate against the host-thread split (reader parse threads / staging threads per run / contexts); one index build, then short runs
cd $GRAFT_REPO_ROOT
for cfg in "8 16 2" "4 8 2" "3 6 2" "2 4 2" "4 8 3" "3 9 3" "4 12 2"; do
  set -- $cfg
  BWAHIP_BENCH_READER_THREADS=$1 BWAHIP_BENCH_HOST_THREADS=$2 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap $3 > gpurun_out/sweep_$1_$2_$3.json 2> gpurun_out/sweep_$1_$2_$3.log || echo "failed $cfg"
  echo "reader $1 host $2 ctx $3: $(grep -E 'file to file' gpurun_out/sweep_$1_$2_$3.log)"
done
