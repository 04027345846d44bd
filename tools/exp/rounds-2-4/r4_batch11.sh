#!/bin/bash
# round 4, GPU batch 11: does it help to keep a buffer's physical chunks from being neighbours?  After a 64 GiB allocation has come and
# gone (the state in which the default backing ran 37 ms), fresh processes: vmm 64 | vmm 64 spread 2 | vmm 64 spread 4 | vmm 1024 |
# vmm 1024 spread 2 | vmm 2 spread 2, with a churn run before each
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b11
mkdir -p $OUT
cd $R
KS=$R/tools/kbench
{
for rep in 1 2 3; do
echo "== round $rep"
for v in "64" "64 --vmm-spread 2" "64 --vmm-spread 4" "1024" "1024 --vmm-spread 2" "2 --vmm-spread 2"; do
  timeout 200 $KS --size 2048 --prec f32 --iters 1 > /dev/null 2>&1      # churn
  echo -n "vmm $v: "
  timeout 150 $KS --size 1024 --prec f64 --iters 8 --vmm $v 2>&1 | grep -E "total passes" | cut -c1-60
done
done
} > $OUT/r4_placement_spread.txt 2>&1
cat $OUT/r4_placement_spread.txt
