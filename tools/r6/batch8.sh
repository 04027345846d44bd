#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/r6/rows_load_probe.py > $O/r6_rows_load_probe.txt 2>&1; grep -v amdgpu.ids $O/r6_rows_load_probe.txt
( DFFT_TEST_SLOW=1 timeout 900 python -m pytest tests/test_gpu_placement.py -m gpu -q -k alloc_free --durations=3 ) > $O/r6_alloc_cycles.txt 2>&1
tail -6 $O/r6_alloc_cycles.txt
DRY=1 T_BENCH=400 T_PROF=400 STEPS=5 WARM=2 bash tools/first_contact.sh 1 fc_dry > $O/r6_first_contact_dry.txt 2>&1
tail -8 $O/r6_first_contact_dry.txt
