#!/bin/bash
# round 4, GPU batch 7: long Bluestein lines (parity), the placement tests, the inverse y pass candidates (batch 6)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b7
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_two_level.py -m gpu -q -x --durations=6 -k "long_bluestein or every_length" > $OUT/r4_pytest_long_bluestein.txt 2>&1
tail -25 $OUT/r4_pytest_long_bluestein.txt
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_multi_device.py tests/test_gpu_slab_sequences.py -m gpu -q --durations=4 > $OUT/r4_pytest_b7.txt 2>&1
tail -8 $OUT/r4_pytest_b7.txt
bash tools/exp/r4_batch6.sh
