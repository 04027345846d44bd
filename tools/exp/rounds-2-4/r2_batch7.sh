#!/bin/bash
# round 2, GPU batch 7: the LDS-free shuffle pass against the LDS configurations (natural lines, fp32), then the round check
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b7
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
echo "=== 512-point lines, fp32, 2 GiB each way: LDS configurations 0 (8.8.8), 6 (32.16 line fastest), 4 (32.16 point fastest); shuffle pass 15 (ds_bpermute), 14 (DPP)"
for v in 0 6 4 15 14; do $K --line 512 --batch 524288 --prec f32 --variant $v --iters 5 --check; done
$K --line 512 --batch 524288 --prec f32 --variant 4 --iters 5 --debug 1
echo "=== 1024-point lines"
for v in 0 6 4 7 15 14; do $K --line 1024 --batch 262144 --prec f32 --variant $v --iters 5 --check; done
} > $OUT/shuffle.txt 2>&1
cat $OUT/shuffle.txt
export TMPDIR=/tmp
for v in 4 15 14; do
( cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_v$v -o v$v -- $K --line 512 --batch 524288 --prec f32 --variant $v --iters 1 > $OUT/pmc_v$v.log 2>&1 )
done
python tools/pmc_summary.py $OUT fft_ > $OUT/shuffle_pmc.txt 2>&1; head -60 $OUT/shuffle_pmc.txt
bash tools/round_check.sh r2
