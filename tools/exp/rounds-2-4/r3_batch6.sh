#!/bin/bash
# round 3, GPU batch 6 (library built with EXTRA=-DDFFT_EXPERIMENTS):
#  (a) C++ drivers + parity tests (the 2048-point real kernels now split / merge in registers)
#  (b) real z passes A/B: Nz = 2048 fp32 R2C radix orders (real_variant 0 = 8.8.16 with 40 B/lane of scratch, 2 = 32.4.8, 3 = 16.16.4,
#      4 = 32.8.4); Nz = 4096 both precisions (real_variant 1 = the round-2 forms through LDS, which spilled)
#  (c) R2C + C2R at 1024^3, both precisions, plain buffers vs tuned placement
#  (d) HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes): fp32 2048^3 multi-rank path, fp32 1024^3 C2C, fp64 1024^3 R2C
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b6
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 900 python -m pytest tests/test_gpu_cpp_drivers.py tests/test_gpu_parity.py tests/test_gpu_placement.py -x -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
{
echo "== fp32 R2C, Nz = 2048 (M = 1024): real_variant 0 | 2 | 3 | 4 | 0"
timeout 200 $K --size 1024x1024x2048 --prec f32 --mode r2c --iters 5 --sweep "real_variant=0;real_variant=2;real_variant=3;real_variant=4;real_variant=0"
for v in 2 3 4; do echo "== check real_variant=$v"; timeout 60 $K --size 256x256x2048 --prec f32 --mode r2c --iters 2 --check --opt real_variant=$v | grep PLAN; done
echo "== fp32 R2C, Nz = 4096 (M = 2048): real_variant 0 (in registers) | 1 (round 2) | 0"
timeout 200 $K --size 512x512x4096 --prec f32 --mode r2c --iters 5 --sweep "real_variant=0;real_variant=1;real_variant=0"
echo "== fp64 R2C, Nz = 4096 (M = 2048): real_variant 0 (in registers) | 1 (round 2) | 0"
timeout 200 $K --size 512x512x4096 --prec f64 --mode r2c --iters 5 --sweep "real_variant=0;real_variant=1;real_variant=0"
} > $OUT/real_ab.txt 2>&1
grep -E "^==|PLAN|z-FFT" $OUT/real_ab.txt | cut -c1-150
{
for p in f64 f32; do
  echo "== R2C + C2R 1024^3 $p, plain hipMalloc buffers"; timeout 100 $K --size 1024 --prec $p --mode r2c --iters 10 --check
  echo "== R2C + C2R 1024^3 $p, tuned placement (4)"; timeout 100 $K --size 1024 --prec $p --mode r2c --iters 10 --check --tune 4
done
echo "== C2C 1024^3 f32, plain"; timeout 100 $K --size 1024 --prec f32 --iters 10
echo "== C2C 1024^3 f32, tuned placement (4)"; timeout 100 $K --size 1024 --prec f32 --iters 10 --tune 4
} > $OUT/r2c_1024.txt 2>&1
grep -E "^==|PLAN|FFT|TUNE|total" $OUT/r2c_1024.txt | cut -c1-200
timeout 400 bash tools/pmc_traffic.sh r3_f32_2048_multirank -- $K --size 2048 --prec f32 --iters 1 --opt mirror_inverse=1 --opt pipeline_chunks=8
timeout 200 bash tools/pmc_traffic.sh r3_f32_1024 -- $K --size 1024 --prec f32 --iters 2
timeout 200 bash tools/pmc_traffic.sh r3_f64_r2c_1024 -- $K --size 1024 --prec f64 --mode r2c --iters 2
python tools/pmc_traffic.py gpurun_out/pmct_r3_f32_2048_multirank $((2*8*2048*2048*2048/8)) "2048^3 fp32 complex, multi-rank code path on one GPU (mirrored inverse, 8 pipeline chunks: 8 launches per pass, tools/kbench --size 2048 --prec f32 --opt mirror_inverse=1 --opt pipeline_chunks=8)" > $OUT/r3_pmc_traffic_f32_2048.json
python tools/pmc_traffic.py gpurun_out/pmct_r3_f32_1024 $((2*8*1024*1024*1024)) "1024^3 fp32 complex, one axis pass per launch (tools/kbench --size 1024 --prec f32)" > $OUT/r3_pmc_traffic_f32_1024.json
python tools/pmc_traffic.py gpurun_out/pmct_r3_f64_r2c_1024 $((2*16*1024*1024*513)) "1024^3 fp64 R2C + C2R (tools/kbench --size 1024 --prec f64 --mode r2c); algorithmic bytes of a y / x pass (513 of 1024 kz planes); the real z passes move 8 B * N^3 + 16 B * N^2 * 513" "dfft::fft_" > $OUT/r3_pmc_traffic_f64_r2c.json
head -c 1500 $OUT/r3_pmc_traffic_f32_2048.json; echo; head -c 700 $OUT/r3_pmc_traffic_f32_1024.json; echo; head -c 2500 $OUT/r3_pmc_traffic_f64_r2c.json
rm -rf gpurun_out/pmct_r3_*/fetch gpurun_out/pmct_r3_*/write 2>/dev/null
