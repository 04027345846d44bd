#!/bin/bash
# round 6, GPU batch 2
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
tools/fp64_peak > $O/r6_fp64_peak.txt 2>&1; cat $O/r6_fp64_peak.txt

# compute_streams (auto = 2 from three chunks on): bit identity, pipelining, relay
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_relay.py -m gpu -q -k "two_compute_streams or pipelined_exchange_chunks or relay" > $O/r6_compute_streams_pytest.txt 2>&1
tail -3 $O/r6_compute_streams_pytest.txt

# alloc / free cycles (incl. the 1 GiB form of the verdict)
( DFFT_TEST_SLOW=1 timeout 900 python -m pytest tests/test_gpu_placement.py -m gpu -q -k "alloc_free_cycles" --durations=5 ) > $O/r6_alloc_cycles.txt 2>&1
tail -6 $O/r6_alloc_cycles.txt

# parity table, complete
rm -f $O/r6_parity_table.txt
( export DFFT_PARITY_TABLE=$PWD/$O/r6_parity_table.txt
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py -m gpu -q \
   -k "fft1d_batched or single_rank_3d_vs_oracle or test_distributed_vs_oracle or r2c_c2r_vs_oracle or c2_256 or c3_512 or c4_1024 or c5_ or 1024_r2c" \
   --durations=15 ) > $O/r6_parity_pytest.txt 2>&1
tail -25 $O/r6_parity_pytest.txt

# the proof: sound and fp32-rounded twiddle tables, both metrics
( export DFFT_AMD_LIBRARY=$PWD/distributedfft_amd/exp/libdfft_amd.so
  DFFT_EXP_F32_TWIDDLES=0 timeout 600 python tools/r6/twiddle_proof.py
  DFFT_EXP_F32_TWIDDLES=1 timeout 600 python tools/r6/twiddle_proof.py ) > $O/r6_f32_twiddle_proof.txt 2>&1
cat $O/r6_f32_twiddle_proof.txt

# R2C rank 0 of 2x4: per-point tables against the segment search (option point_tables), one compute stream so that spans add up
{
for pt in 1 0; do
  timeout 300 tools/kbench --size 1024 --prec f64 --mode r2c --ranks 2x4 --rank 0 --iters 10 --lib-buffers --opt compute_streams=1 --opt point_tables=$pt
done
timeout 300 tools/kbench --size 1024 --prec f64 --mode r2c --ranks 2x4 --rank 0 --iters 10 --lib-buffers --opt compute_streams=1 --opt uniform_tables=0
} > $O/r6_r2c_tables_ab.txt 2>&1
grep -E "PLAN|FFT|total" $O/r6_r2c_tables_ab.txt

DRY=1 T_BENCH=400 T_PROF=400 STEPS=5 WARM=2 bash tools/first_contact.sh 1 fc_dry > $O/r6_first_contact_dry.txt 2>&1
tail -15 $O/r6_first_contact_dry.txt
