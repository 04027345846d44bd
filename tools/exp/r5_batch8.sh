#!/bin/bash
# round 5, GPU batch 8: rank 0 of the 2x4 plan, execR2C + execC2R at 1024^3 fp64 (the reference's own API on the BASELINE grid): HBM traffic per
# kernel from PMC counters (reference layout and spectral_layout = 1) and the rocprofv3 kernel summary
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b8
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
K="$R/tools/kbench --size 1024 --prec f64 --iters 2 --lib-buffers --ranks 2x4 --rank 0 --mode r2c"
# algorithmic bytes of one pass of this rank: the Hermitian half [1024][512][129] read once and written once
ALG=$((2 * 16 * 1024 * 512 * 129))
bash tools/pmc_traffic.sh r5_r2c_rank0 -- $K > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_r5_r2c_rank0 $ALG "rank 0 of 2x4, 1024^3 fp64 execR2C + execC2R, 4 chunk launches per pass: per LAUNCH a quarter of 2 x 16 B x 1024 x 512 x 129 (the real z passes move 8 B reals on one side)" "dfft::fft_" > $OUT/r5_pmc_traffic_f64_r2c_rank0_2x4.json 2>&1
bash tools/pmc_traffic.sh r5_r2c_rank0_sp -- $K --opt spectral_layout=1 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_r5_r2c_rank0_sp $ALG "the same with spectral_layout = 1" "dfft::fft_" > $OUT/r5_pmc_traffic_f64_r2c_rank0_2x4_spectral.json 2>&1
rm -rf $R/gpurun_out/pmct_*
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b8")
for f in ("r5_pmc_traffic_f64_r2c_rank0_2x4.json", "r5_pmc_traffic_f64_r2c_rank0_2x4_spectral.json"):
    try:
        j = json.load(open(os.path.join(d, f)))
        print(f, "launches", j["dispatches"], "avg bytes per launch", round(j["hbm_bytes_per_launch"] / 1e6, 1), "MB; a quarter pass is", round(j["algorithmic_bytes_per_launch"] / 4 / 1e6, 1), "MB")
        for k, v in j["per_kernel"].items():
            print("   ", k[:110], v["launches"], round(v["read_bytes_per_launch"] / 1e6, 1), round(v["write_bytes_per_launch"] / 1e6, 1))
    except Exception as e:
        print(f, "unreadable", e)
PY
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r2c -- $R/tools/kbench --size 1024 --prec f64 --iters 20 --lib-buffers --ranks 2x4 --rank 0 --mode r2c > $OUT/prof.log 2>&1 )
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/r5_rank0_2x4_f64_r2c_kernel_stats.csv \;
rm -rf $OUT/prof
head -12 $OUT/r5_rank0_2x4_f64_r2c_kernel_stats.csv | cut -c1-220
