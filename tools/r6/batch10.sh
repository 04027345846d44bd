#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cpp_drivers.py tests/test_gpu_cli.py tests/test_gpu_slab_sequences.py tests/test_gpu_spectral.py tests/test_gpu_cpp_shim.py -m gpu -q --durations=12 -k "testcase4 or laplacian or cli or spectrum or z_then_yx or one_process" > $O/r6_batch10_pytest.txt 2>&1
tail -30 $O/r6_batch10_pytest.txt
