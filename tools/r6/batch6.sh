#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
{
echo "== mode hint: address ranges go back to the runtime (memory is returned) but every reservation asks for a never-used address"
timeout 200 tools/vmm_reuse_repro --mode hint --use copy --threads 0 --grow 0 --seconds 12 --report 2000
timeout 200 tools/vmm_reuse_repro --mode hint --use copy --threads 8 --pull 1 --seconds 15
timeout 200 tools/vmm_reuse_repro --mode hint --use kernel --threads 8 --pull 1 --seconds 15
timeout 200 tools/vmm_reuse_repro --mode hint --use copy --threads 8 --pull 1 --seconds 15 --mib 1024 --chunk-mib 1024
timeout 200 tools/vmm_reuse_repro --mode hint --hint-base-tib 64 --use kernel --threads 8 --pull 0 --seconds 15 --mib 256 --chunk-mib 64
} > $O/r6_vmm_hint.txt 2>&1
cat $O/r6_vmm_hint.txt
