#!/bin/bash
# tools/pmc_traffic.sh TAG -- <command...>: FETCH_SIZE and WRITE_SIZE in two separate PMC passes (kernel trace only)
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmct_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD=("$@")
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- "${CMD[@]}" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- "${CMD[@]}" > $OUT/write.log 2>&1
