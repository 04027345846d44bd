#!/bin/bash
# round 3, GPU batch 13: two-level lines (tests/test_gpu_two_level.py) and every test of the generic (Bluestein) kernel after its restructuring
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b13
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_two_level.py -x -q --durations=15 > $OUT/pytest_two_level.txt 2>&1; tail -40 $OUT/pytest_two_level.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_sequences.py tests/test_gpu_round3.py -x -q -k "any_length or any_size or bluestein or y_then_zx or long or error or randomised" > $OUT/pytest_generic.txt 2>&1; tail -5 $OUT/pytest_generic.txt
