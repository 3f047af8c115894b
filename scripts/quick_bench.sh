#!/bin/bash
# GPU tests + a short 128 Mbp bench; prints one summary line.  usage: gpurun -- scripts/quick_bench.sh [notest]
cd $GRAFT_REPO_ROOT
if [ "$1" != "notest" ]; then timeout -k 5 400 python -m pytest tests -m gpu -q -x > gpurun_out/t6.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t6.log; fi
timeout -k 10 200 python bench.py --genome-mbp 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.log; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/bench_q.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['tail_us'])"
