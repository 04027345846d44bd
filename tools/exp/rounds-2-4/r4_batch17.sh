#!/bin/bash
# round 4, GPU batch 17: reduced check of the allocator change (buffers built from chunks far apart; kernels unchanged): PMC traffic of the
# new library (sha256), placement / fullsize parity subset + smoke, bench lines from fresh processes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b17
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
TAG=r4d
sha256sum distributedfft_amd/libdfft_amd.so > $OUT/${TAG}_library_sha256.txt
bash tools/pmc_traffic.sh ${TAG}_f64_1024 -- $R/tools/kbench --size 1024 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1024 34359738368 "1024^3 fp64 complex, one axis pass per launch (tools/kbench --size 1024 --prec f64)" > $OUT/${TAG}_pmc_traffic.json 2>&1
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
rm -rf $R/gpurun_out/pmct_*
python -c "import json; j=json.load(open('$OUT/${TAG}_pmc_traffic.json')); print('traffic/alg', round(j['hbm_bytes_per_launch']/j['algorithmic_bytes_per_launch'],4), j['library_sha256'][:12])"
timeout 400 python -m pytest tests/test_gpu_placement.py tests/test_gpu_fullsize.py -m gpu -q -x -k "placement or malloc or tune or c2_256 or c3_512 or 1024_r2c" > $OUT/${TAG}_pytest_subset.txt 2>&1; tail -3 $OUT/${TAG}_pytest_subset.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_${TAG}.json 2> $OUT/e.err
for i in 1 2; do timeout 200 python bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_${TAG}_$i.json 2>> $OUT/e.err; done
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b17")
for f in sorted(glob.glob(os.path.join(d, "bench_r4d*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); c = j["config"]
        print(os.path.basename(f), j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("traffic"), c["placement"].get("alloc_seconds_out_and_back"), {k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k}, c.get("plain_buffers_ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
PY
