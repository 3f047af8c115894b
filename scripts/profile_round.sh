#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench command, PMC passes (own runs), then the default bench line
# itself (last, so that it can quote the traffic of these very PMC passes).
#   gpurun --timeout 1100 -- scripts/profile_round.sh <tag> <round-dir>
set -e
TAG=${1:-prof}; ROUND=${2:-profiles/r01}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log || echo "trace failed"
echo "trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed"
  echo "pmc $i done"
done
cd $GRAFT_REPO_ROOT
touch $OUT/bench_default.json
python3 scripts/collect_profiles.py $TAG $ROUND --traffic-only > /dev/null
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
echo "bench done"; tail -c 700 $OUT/bench_default.json
