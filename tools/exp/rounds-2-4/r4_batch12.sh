#!/bin/bash
# round 4, GPU batch 12: 1 GiB chunks, every 2nd / 3rd created chunk kept, against plain 1 GiB chunks: more fresh processes, with and
# without a 64 GiB allocation coming and going in between
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b12
mkdir -p $OUT
cd $R
KS=$R/tools/kbench
{
for rep in 1 2 3 4; do
echo "== round $rep$( [ $((rep % 2)) = 0 ] && echo ' (churn before every run)')"
for v in "1024" "1024 --vmm-spread 2" "1024 --vmm-spread 3"; do
  if [ $((rep % 2)) = 0 ]; then timeout 200 $KS --size 2048 --prec f32 --iters 1 > /dev/null 2>&1; fi
  echo -n "vmm $v: "
  timeout 150 $KS --size 1024 --prec f64 --iters 8 --vmm $v 2>&1 | grep -E "FFT|total passes" | awk '{printf "%s %s  ", $1, $3} END {print ""}'
done
done
} > $OUT/r4_placement_spread_1gib.txt 2>&1
cat $OUT/r4_placement_spread_1gib.txt
