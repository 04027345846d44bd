#!/bin/bash
# tools/first_contact.sh [GPUS] [TAG] -- the first thing to run on a node with more than one MI355X (round-5 verdict, item 7).
# Nothing multi-GPU in this repository has met real xGMI links: every exchange time in DESIGN.md section 5 is a link-rate model
# (bench.py xgmi_model: bytes per link / 153 GB/s).  This script produces, in one go and each step under its own timeout,
#   1. tools/rccl_smoke.py --gpus N                    RCCL version, per-link / all-to-all-v / schedule rates of the native transport
#   2. bench.py --gpus 2 / 4 / 8 (up to N)             the headline line per GPU count; on pencil grids bench.py itself times the
#      --transport rccl and --transport torch          direct and the relayed exchange (config.direct / config.relay)
#   3. a rocprofv3 kernel trace of the N-GPU run       (rank 0's process; --kernel-trace --stats only: no PMC next to a trace)
#   4. a table: measured exchange spans against the xGMI model, per exchange and GPU count
# and leaves everything under gpurun_out/first_contact/ (copy what is to be kept into profiles/).
#
#   DRY=1 tools/first_contact.sh 1          syntax / plumbing check on one GPU: the same commands with --gpus 1, and the 8-rank
#                                           world as 8 gloo ranks sharing the GPU (what tests/test_gpu_round3.py does for bench.py)
set -u
N=${1:-8}
TAG=${2:-fc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/first_contact
mkdir -p "$OUT"
cd "$R" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T_SMOKE=${T_SMOKE:-300}; T_BENCH=${T_BENCH:-900}; T_PROF=${T_PROF:-900}
STEPS=${STEPS:-10}; WARM=${WARM:-3}
log() { echo "[first_contact $(date -u +%H:%M:%S)] $*" | tee -a "$OUT/${TAG}_log.txt"; }
run() {   # run NAME TIMEOUT cmd...   -> $OUT/NAME.json (stdout), $OUT/NAME.err (stderr); never aborts the script
    local name=$1 to=$2; shift 2
    log "$name: $* (timeout $to s)"
    local t0; t0=$(date +%s)
    timeout "$to" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    local rc=$?
    log "$name: exit $rc after $(( $(date +%s) - t0 )) s"
    return 0
}

python - <<'PY' | tee -a "$OUT/${TAG}_log.txt"
import torch
print("devices:", torch.cuda.device_count(), [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())][:1])
PY
rocm-smi --showtopo > "$OUT/${TAG}_topology.txt" 2>&1 || true

# 1. the transport alone
run ${TAG}_rccl_smoke_${N}gpu $T_SMOKE python tools/rccl_smoke.py --gpus "$N" --mib 256 --iters 5

# 2. the headline per GPU count and transport
COUNTS="2 4 8"
[ "${DRY:-0}" = 1 ] && COUNTS="1"
for g in $COUNTS; do
    [ "$g" -gt "$N" ] && [ "${DRY:-0}" != 1 ] && continue
    for tr in rccl torch; do
        run ${TAG}_bench_${g}gpu_${tr} $T_BENCH python bench.py --gpus "$g" --steps "$STEPS" --warmup "$WARM" --transport $tr --no-cpu-baseline
    done
done
if [ "${DRY:-0}" = 1 ]; then
    # the 8-rank world on ONE GPU: gloo ranks sharing the device (no RCCL: it refuses two ranks per device), small grid
    run ${TAG}_bench_8ranks_gloo_1gpu $T_BENCH python bench.py --gpus 8 --backend gloo --transport torch --size 256 --steps 3 --warmup 1 --no-cpu-baseline
fi

# 3. kernel trace of the largest run (rank 0 of the self-launched world is the profiled process' child: trace the launcher tree)
G=$N; [ "${DRY:-0}" = 1 ] && G=1
( cd /tmp && timeout "$T_PROF" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${G}gpu" -o bench -- \
    python "$R/bench.py" --gpus "$G" --steps 5 --warmup 2 --transport rccl --no-cpu-baseline --relay 0 --no-multi-rank-path --no-plain-leg \
    > "$OUT/${TAG}_bench_${G}gpu_profiled.json" 2> "$OUT/${TAG}_bench_${G}gpu_profiled.err" )
log "profiled run: exit $?"
find "$OUT/prof_${G}gpu" -name "*kernel_stats.csv" | head -8 | while read -r f; do cp "$f" "$OUT/${TAG}_${G}gpu_$(basename "$(dirname "$f")")_kernel_stats.csv"; done
rm -rf "$OUT/prof_${G}gpu"

# 4. measured against the model
python - "$OUT" "$TAG" <<'PY' | tee "$OUT/${TAG}_table.txt"
import glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
print(f"{'run':34s} {'ms/step':>9s} {'GFLOP/s':>10s} {'hidden':>7s}  per exchange and transform: measured ms (GB/s per link) | model ms at 153 GB/s per link")
for f in sorted(glob.glob(os.path.join(out, f"{tag}_bench_*.json"))):
    line = None
    for ln in open(f):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    name = os.path.basename(f)[len(tag) + 1:-5]
    if line is None:
        print(f"{name:34s}  no JSON line (see {os.path.basename(f)[:-5]}.err)")
        continue
    j = json.loads(line)

    def row(label, d):
        ex = (d.get("xgmi") or {}).get("per_exchange_per_transform") or {}
        cells = "; ".join(f"{k}: {v.get('measured_ms')} ({v.get('measured_GBps_per_link')}) | {v.get('predicted_ms')}" for k, v in ex.items())
        hid = (d.get("overlap") or {}).get("hidden_frac")
        print(f"{label:34s} {d.get('ms_per_step', float('nan')):9.3f} {d.get('value', float('nan')) if 'value' in d else float('nan'):10.1f} {str(hid):>7s}  {cells}")
    row(name, j)
    for leg in ("direct", "relay", "alt"):
        if isinstance(j.get("config", {}).get(leg), dict):
            row("  config." + leg, j["config"][leg])
PY
log "done: $(ls "$OUT" | wc -l) files in $OUT"
