#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
DFFT_TEST_SLOW=1 timeout 1500 python -m pytest tests/test_gpu_cpp_drivers.py -m gpu -q --durations=6 -k "largest" > $O/r6_batch11_pytest.txt 2>&1
tail -25 $O/r6_batch11_pytest.txt
