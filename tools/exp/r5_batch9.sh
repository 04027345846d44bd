#!/bin/bash
# round 5, GPU batch 9: dfft_free retires virtual address ranges instead of returning them: the relay stress with virtual-memory staging
# (must now be clean), with the old behaviour next to it; PMC traffic of the final library; placement / relay tests; two bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b9
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{
echo "== staging = virtual memory, address ranges retired (the default now)"; DFFT_RELAY_STAGING=vmm python tools/exp/r5_relay_stress.py 8 2>&1 | grep "66, 50\|chunks=3" | grep -v "direct vs"
echo "== staging = virtual memory, DFFT_VMM_RETIRE_TIB=0 (ranges returned to the runtime: the behaviour before)"; DFFT_RELAY_STAGING=vmm DFFT_VMM_RETIRE_TIB=0 python tools/exp/r5_relay_stress.py 8 2>&1 | grep "66, 50" | grep -v "direct vs"
} > $OUT/stress_retire.txt 2>&1
cat $OUT/stress_retire.txt
TAG=r5f
sha256sum distributedfft_amd/libdfft_amd.so > $OUT/${TAG}_library_sha256.txt
bash tools/pmc_traffic.sh ${TAG}_f64_1024 -- $R/tools/kbench --size 1024 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1024 34359738368 "1024^3 fp64 complex, one axis pass per launch (tools/kbench --size 1024 --prec f64)" > $OUT/${TAG}_pmc_traffic.json 2>&1
rm -rf $R/gpurun_out/pmct_*
python -c "import json; j=json.load(open('$OUT/${TAG}_pmc_traffic.json')); print('traffic/alg', round(j['hbm_bytes_per_launch']/j['algorithmic_bytes_per_launch'],4), j['library_sha256'][:12])"
timeout 400 python -m pytest tests/test_gpu_placement.py tests/test_gpu_relay.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_${TAG}_$i.json 2>> $OUT/bench.err; done
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b9")
for f in sorted(glob.glob(os.path.join(d, "bench_r5f_*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); c = j["config"]
        print(os.path.basename(f), j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("traffic"), j["round_trip_rel_linf"], {k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
