#!/bin/bash
# round 5, GPU batch 7: the allocator with the 0.95 threshold (dfft.o only: kernels unchanged): placement tests, PMC traffic of the new
# library (sha256), bench lines from three fresh processes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b7
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
TAG=r5e
sha256sum distributedfft_amd/libdfft_amd.so > $OUT/${TAG}_library_sha256.txt
bash tools/pmc_traffic.sh ${TAG}_f64_1024 -- $R/tools/kbench --size 1024 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1024 34359738368 "1024^3 fp64 complex, one axis pass per launch (tools/kbench --size 1024 --prec f64)" > $OUT/${TAG}_pmc_traffic.json 2>&1
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
rm -rf $R/gpurun_out/pmct_*
python -c "import json; j=json.load(open('$OUT/${TAG}_pmc_traffic.json')); print('traffic/alg', round(j['hbm_bytes_per_launch']/j['algorithmic_bytes_per_launch'],4), j['library_sha256'][:12])"
timeout 300 python -m pytest tests/test_gpu_placement.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_${TAG}_$i.json 2>> $OUT/bench.err; done
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b7")
for f in sorted(glob.glob(os.path.join(d, "bench_r5e_*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); c = j["config"]; pl = c["placement"]
        print(os.path.basename(f), j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("traffic"), pl.get("alloc_seconds_out_and_back"),
              {k: (pl[k]["kept"][:26], pl[k]["probe_TBps"], pl[k]["good_threshold_TBps"], pl[k]["candidates_drawn"], pl[k]["seconds"]) for k in ("out", "back") if k in pl},
              {k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
