#!/bin/bash
# tools/round_check.sh -- the standard measurement set of a round in ONE gpurun call (≈ 3 GPU-minutes):
#   gpurun --timeout 900 -- 'bash tools/round_check.sh'
# GPU parity suite, bench line, rocprofv3 kernel stats of the bench, per-pass times (C2C fp64/fp32,
# R2C fp64/fp32, chunked), R2C wall times.  Everything lands in gpurun_out/round_check/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_check
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -2 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; echo
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/prof.log 2>&1 )
cp $OUT/prof/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
{
  echo "== c2c fp64";            python tools/phase_times.py 1024 double 3 | tail -7
  echo "== c2c fp32";            python tools/phase_times.py 1024 float 3 | tail -7
  echo "== c2c fp64 chunks 8";   DFFT_CHUNKS=8 python tools/phase_times.py 1024 double 3 | tail -7
  echo "== c2c fp64 chunks 32";  DFFT_CHUNKS=32 python tools/phase_times.py 1024 double 3 | tail -7
  echo "== r2c fp64";            python tools/phase_times_r2c.py 1024 double 3
  echo "== r2c fp32";            python tools/phase_times_r2c.py 1024 float 3
  echo "== r2c wall";            python tools/latency.py 2>&1 | grep R2C
} > $OUT/phase_times.txt 2>&1
grep -v amdgpu.ids $OUT/phase_times.txt
