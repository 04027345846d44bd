#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/r6/rows_load_probe.py > $O/r6_rows_load_probe.txt 2>&1; cat $O/r6_rows_load_probe.txt
timeout 1500 python -m pytest tests/test_gpu_two_level.py tests/test_gpu_relay.py tests/test_gpu_slab_sequences.py tests/test_gpu_round3.py tests/test_gpu_parity.py -q -m gpu --durations=0 -k "not c4_1024 and not c5_ and not bench_multi" > $O/r6_durations.txt 2>&1
grep -E "s call" $O/r6_durations.txt | awk '{split($3,a,"::"); split(a[2],b,"["); t[b[1]]+=$1; n[b[1]]++} END {for (k in t) printf "%8.1f s %4d  %s\n", t[k], n[k], k}' | sort -rn | head -40
tail -3 $O/r6_durations.txt
DRY=1 T_BENCH=400 T_PROF=400 STEPS=5 WARM=2 bash tools/first_contact.sh 1 fc_dry > $O/r6_first_contact_dry.txt 2>&1
tail -8 $O/r6_first_contact_dry.txt
