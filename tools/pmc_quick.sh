#!/bin/bash
# tools/pmc_quick.sh TAG -- <command...>: two PMC passes (LDS, activity), kernel-trace only
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD=("$@")
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/lds -o lds -- "${CMD[@]}" > $OUT/lds.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/act -o act -- "${CMD[@]}" > $OUT/act.log 2>&1
