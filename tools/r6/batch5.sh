#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
{
echo "== is the physical memory of an unmapped range returned?  free memory (hipMemGetInfo) every 1000 cycles of 64 MiB"
for mode in retire free arena; do for late in 0 1; do
  echo "-- mode $mode, hipMemRelease $( [ $late = 1 ] && echo 'after hipMemUnmap' || echo 'right after hipMemMap' )"
  timeout 200 tools/vmm_reuse_repro --mode $mode --arena-gib 2048 --use copy --threads 0 --grow 0 --seconds 12 --report 1000 --release-late $late
done; done
echo "== correctness of the late release in the modes that never reuse an address, other threads enqueueing"
timeout 200 tools/vmm_reuse_repro --mode retire --use copy --threads 8 --pull 1 --seconds 12 --release-late 1
timeout 200 tools/vmm_reuse_repro --mode arena --arena-gib 1024 --use kernel --threads 8 --pull 1 --seconds 12 --release-late 1
timeout 200 tools/vmm_reuse_repro --mode free --use copy --threads 8 --pull 1 --seconds 12 --release-late 1
} > $O/r6_vmm_cost.txt 2>&1
cat $O/r6_vmm_cost.txt
