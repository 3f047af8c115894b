#!/bin/bash
# Round profile, two gpurun calls (one call is capped at 20 minutes and the 3.1 Gbp index is rebuilt on every fresh box):
#   gpurun --timeout 1200 -- scripts/profile_round.sh counters <tag>   rocprofv3 kernel stats of the bench command's one-batch-at-a-time region
#                                                                      (--overlap 1: the region kernel_ms and roofline come from), then the PMC
#                                                                      passes (own runs, 2 M reads = 2 launches of every kernel per pass)
#   scripts/collect_profiles.py <tag> profiles/rNN                    here: summaries -> profiles/rNN, profiles/pmc_traffic.json
#   gpurun --timeout 1200 -- scripts/profile_round.sh bench <tag>      the default bench line itself (quotes the traffic of those PMC passes)
set -e
MODE=${1:-counters}; TAG=${2:-prof}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
if [ "$MODE" = counters ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --overlap 1 --no-cpu-baseline --no-e2e > $OUT/bench_under_trace.json 2> $OUT/trace.log || echo "trace failed"
  echo "trace done"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --reads 2000000 --overlap 1 --no-cpu-baseline --no-e2e > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed"
    echo "pmc $i done"
  done
else
  cd $GRAFT_REPO_ROOT
  timeout -k 10 900 python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $OUT/bench_default.json 2> $OUT/bench_default.log
  echo "bench done"; tail -c 900 $OUT/bench_default.json
fi
