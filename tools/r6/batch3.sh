#!/bin/bash
# round 6, GPU batch 3: the whole GPU suite on the split library (+ the slow remainder), first-contact dry run, one bench line
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -f $O/r6_parity_table.txt
( export DFFT_PARITY_TABLE=$PWD/$O/r6_parity_table.txt
  timeout 1800 python -m pytest tests -q -m gpu --durations=30 ) > $O/r6_pytest_gpu.txt 2>&1
tail -45 $O/r6_pytest_gpu.txt
timeout 600 python -m pytest tests -q -m "gpu and slow" --durations=5 > $O/r6_pytest_gpu_slow.txt 2>&1
tail -8 $O/r6_pytest_gpu_slow.txt
DRY=1 T_BENCH=400 T_PROF=400 STEPS=5 WARM=2 bash tools/first_contact.sh 1 fc_dry > $O/r6_first_contact_dry.txt 2>&1
tail -12 $O/r6_first_contact_dry.txt
timeout 500 python bench.py > $O/bench_r6a.json 2> $O/bench_r6a.err; tail -c 1500 $O/bench_r6a.json; tail -3 $O/bench_r6a.err
