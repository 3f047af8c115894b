#!/bin/bash
# PMC passes for k_smem (separate runs, counters only).  usage: scripts/pmc_smem.sh <lanes> <outdir>
set -e
G=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$2; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export BWAHIP_SMEM_LANES=$G
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --genome-mbp 128 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
