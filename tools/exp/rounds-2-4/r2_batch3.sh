#!/bin/bash
# round 2, GPU batch 3: is the bimodal speed of the strided-128-B passes a property of the physical backing of a buffer?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b3
mkdir -p $OUT
cd $R
{
for i in 1 2; do echo "=== memprobe run $i"; $R/tools/memprobe 16 3; done
echo "=== repeated identical kbench runs (process-to-process variation)"
for i in 1 2 3 4; do
  $R/tools/kbench --size 1024 --prec f64 --iters 3 --label rep$i-default
  $R/tools/kbench --size 1024 --prec f64 --iters 3 --label rep$i-ntall --opt variant_fz=3 --opt variant_fy=3 --opt variant_fx=3
done
} > $OUT/probe.txt 2>&1
tail -5 $OUT/probe.txt
