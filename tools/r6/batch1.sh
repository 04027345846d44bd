#!/bin/bash
# round 6, GPU batch 1: (a) the standalone virtual-memory address-reuse repro, (b) the per-entry parity table on the BASELINE
# configurations, (c) proof that the per-entry bound catches a twiddle table rounded to fp32 (A/B library only).
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tools/vmm_reuse_repro (standalone, no libdfft), $(date -u)"
for use in copy kernel; do for mode in free keep retire; do for pull in 0 1; do
  timeout 120 tools/vmm_reuse_repro --mode $mode --use $use --pull $pull --threads 8 --seconds 12 --mib 64 --chunk-mib 2; echo "   exit $?"
done; done; done
echo "== 1 GiB chunks, like the library's default backing"
for mode in free keep; do
  timeout 120 tools/vmm_reuse_repro --mode $mode --use copy --pull 1 --threads 8 --seconds 12 --mib 1024 --chunk-mib 1024; echo "   exit $?"
done
echo "== no other thread enqueueing"
timeout 120 tools/vmm_reuse_repro --mode free --use copy --pull 0 --threads 0 --seconds 10; echo "   exit $?"
} > $O/r6_vmm_reuse_repro.txt 2>&1

rm -f $O/r6_parity_table.txt
export DFFT_PARITY_TABLE=$PWD/$O/r6_parity_table.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py -m gpu -q \
   -k "fft1d_batched or single_rank_3d_vs_oracle or test_distributed_vs_oracle or r2c_c2r_vs_oracle or c2_256 or c3_512 or c4_1024 or c5_ or 1024_r2c" \
   --durations=15 > $O/r6_parity_pytest.txt 2>&1
echo "parity pytest exit $?" >> $O/r6_parity_pytest.txt
unset DFFT_PARITY_TABLE

if [ -f distributedfft_amd/exp/libdfft_amd.so ]; then
{
echo "== fp64 forward checks on the A/B library with the twiddle table rounded to fp32 (DFFT_EXP_F32_TWIDDLES=1): they must FAIL on the per-entry bound"
export DFFT_AMD_LIBRARY=$PWD/distributedfft_amd/exp/libdfft_amd.so
for t in "tests/test_gpu_parity.py::test_fft1d_batched_vs_oracle[1024-double]" "tests/test_gpu_parity.py::test_single_rank_3d_vs_oracle[shape6-double]" \
         "tests/test_gpu_fullsize.py::test_c2_256_single_gpu_every_point_vs_oracle[False]" "tests/test_gpu_fullsize.py::test_c2_256_single_gpu_every_point_vs_oracle[True]"; do
  echo "---- $t : sound table"
  DFFT_EXP_F32_TWIDDLES=0 timeout 300 python -m pytest "$t" -q 2>&1 | tail -2
  echo "---- $t : fp32-rounded table"
  DFFT_EXP_F32_TWIDDLES=1 timeout 300 python -m pytest "$t" -q 2>&1 | grep -E "AssertionError|assert |passed|failed" | head -6
done
} > $O/r6_f32_twiddle_proof.txt 2>&1
fi
tail -5 $O/r6_parity_pytest.txt

# (d) compute_streams = 2: bit identity, then the whole-step time of rank 0's plans of the 8-GPU grid (exchange stubbed) by depth
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "two_compute_streams or pipelined_exchange_chunks" > $O/r6_compute_streams_pytest.txt 2>&1
tail -3 $O/r6_compute_streams_pytest.txt
{
echo "== rank 0 of 2x4, 1024^3 fp64, exchange stubbed: forward + inverse back to back, no phase events (tools/kbench --wall-only)"
for mode in c2c r2c; do
  for ch in 1 2 4 8; do
    for cs in 1 2; do
      [ $ch = 1 ] && [ $cs = 2 ] && continue
      timeout 300 tools/kbench --size 1024 --prec f64 --mode $mode --ranks 2x4 --rank 0 --iters 20 --lib-buffers --tune-variants --wall-only --opt pipeline_chunks=$ch --opt compute_streams=$cs | grep WALL
    done
  done
done
echo "== fp32 2048^3 (C5), rank 0 of 2x4"
for ch in 1 4; do for cs in 1 2; do
  [ $ch = 1 ] && [ $cs = 2 ] && continue
  timeout 300 tools/kbench --size 2048 --prec f32 --mode c2c --ranks 2x4 --rank 0 --iters 10 --lib-buffers --tune-variants --wall-only --opt pipeline_chunks=$ch --opt compute_streams=$cs | grep WALL
done; done
} > $O/r6_compute_streams.txt 2>&1
cat $O/r6_compute_streams.txt
