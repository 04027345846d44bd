#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
( DFFT_TEST_SLOW=1 timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_relay.py -m gpu -q --durations=8 ) > $O/r6_alloc_cycles.txt 2>&1
tail -14 $O/r6_alloc_cycles.txt
echo "== relay stress with virtual-memory staging (the configuration that failed 8/8 in round 5 with ranges returned)" > $O/r6_relay_stress.txt
DFFT_RELAY_STAGING=vmm timeout 600 python tools/exp/r5_relay_stress.py 8 >> $O/r6_relay_stress.txt 2>&1
tail -12 $O/r6_relay_stress.txt
( timeout 1800 python -m pytest tests -q -m gpu --durations=25 ) > $O/r6_pytest_gpu.txt 2>&1
tail -32 $O/r6_pytest_gpu.txt
