#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
( export DFFT_PARITY_TABLE=$PWD/$O/r6_parity_table_c5.txt; rm -f $DFFT_PARITY_TABLE
  timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_relay.py tests/test_gpu_two_level.py tests/test_gpu_slab_sequences.py tests/test_gpu_cpp_drivers.py tests/test_gpu_cpp_shim.py -m gpu -q --durations=8 -k "c5_2048 or relay or staging or two_level or long or slab or pencil" ) > $O/r6_batch9_pytest.txt 2>&1
tail -16 $O/r6_batch9_pytest.txt; cat $O/r6_parity_table_c5.txt | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
