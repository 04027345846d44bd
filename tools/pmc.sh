#!/bin/bash
# tools/pmc.sh TAG -- <command...>
# Collects rocprofv3 PMC counters for a command in separate passes (one counter group per run,
# kernel-trace only, as the MI355X guide prescribes) and a --stats kernel trace.
# Output: gpurun_out/pmc_TAG/<group>/*.csv
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- "${CMD[@]}" > $OUT/$name.log 2>&1
}
CMD=("$@")
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- "${CMD[@]}" > $OUT/stats.log 2>&1
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
find $OUT -name "*.csv" | head -40
