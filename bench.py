#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): 3-D FFT GFLOP/s with the 5*N^3*log2(N^3) convention, forward + inverse,
1024^3 fp64 complex, at 1/2/4/8 GPUs.  A "step" is one forward plus one inverse transform of the
resident grid (the reference's testcase 0 + testcase 2 back to back,
tests/src/pencil/random_dist_3D.cu:154-227, :506-579), inputs already in HBM.  Protocol of the
reference's jobs: 10 warm-up + 20 timed iterations (jobs/bwunicluster/pencil/benchmarks_base.json:4-8).

    python bench.py                      # 1 GPU, 1024^3 fp64, 10 warm-up + 20 steps
    python bench.py --size 2048 --precision float          # config 5's grid on one GPU (in = back aliased)
    python bench.py --gpus 8             # starts its own 8 ranks (torch.distributed.run on 127.0.0.1, a free port) ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8        # ... or runs inside ranks somebody else started
    python bench.py --dry-run            # build the C4 / C5 plans of all 8 ranks on a host without a GPU

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline          -- achieved HBM GB/s of the dominant kernel (the axis-pass kernel), from algorithmic bytes per
                       launch / HIP-event duration on the launch stream
  cpu_baseline      -- the CPU oracle (a port: the reference has no CPU path) timed on this host
  config.per_pass   -- device time and achieved TB/s of every pass of the timed plan
  config.multi_rank_path (N = 1 only) -- the same grid through the code path every rank runs at N > 1: mirrored
                       inverse order (x, y, z) and segmented (8-chunk) address tables, so that a 1-GPU lease
                       predicts the per-GPU compute of the multi-GPU runs
  config.per_gpu_kernels_8gpu (N = 1 only) -- rank 0's plan of the 8-GPU decompositions (pencil 2x4 = BASELINE C4, slab 8)
                       executed on this GPU with the exchange stubbed out: the kernels one GPU of the 8-GPU run launches
                       (its 1/8 of the volume, its segment tables and pipeline chunks), timed with HIP events.  Both legs
                       call the plan's own tuner first (dfft_tune_variants, as bench.py does at N > 1; `tune_variants`
                       holds the as-built and the chosen time; --no-tune-variants measures the rule-based plan)
  xgmi (N > 1)      -- bytes per link and the time the links need at the guide's 153 GB/s (predicted_ms) next to the
                       measured exchange spans; overlap.hidden_frac = 1 - (step - sum kernels) / sum exchanges

At N > 1 the headline decomposition is the one BASELINE.json names (8 GPUs: pencil 2x4, 4: pencil 2x2, 2: slab); the slab
decomposition over all N ranks -- one exchange over N-1 private xGMI links instead of two over 3 + 1 -- is measured in the
same run and reported as config.alt (--decomp slab makes it the headline).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_PEAK_GBS = 6290.0   # same guide: what a plain device-to-device copy reaches (read + write bytes / time)
XGMI_LINK_GBS = 153.0   # same guide: per link and direction
FP64_VECTOR_PEAK_TFLOPS = 73.4   # measured on an MI355X: tools/fp64_peak (profiles/r6_fp64_peak.txt); public specification 78.6
FP32_VECTOR_PEAK_TFLOPS = 157.3  # public specification (packed FMA); not measured here
RELAY_TIMEOUT_S = 180   # watchdog of the relayed leg at N > 1 (it has never run on more than one GPU)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=1024, help="cube edge (BASELINE: 1024)")
    ap.add_argument("--precision", default="double", choices=["double", "float"])
    ap.add_argument("--p1", type=int, default=0)
    ap.add_argument("--p2", type=int, default=0)
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "torch"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU (functional test)")
    ap.add_argument("--decomp", default="auto", choices=["auto", "slab", "pencil"],
                    help="auto = pencil = the BASELINE.json grids (2x4 at 8 GPUs, 2x2 at 4, slab at 2); slab = one exchange over all ranks")
    ap.add_argument("--no-dup-channel", action="store_true",
                    help="native RCCL transport: do not duplicate the communicator for the second exchange of a pencil plan")
    ap.add_argument("--no-alt", action="store_true", help="skip the alternative-decomposition measurement")
    ap.add_argument("--no-multi-rank-path", action="store_true", help="N = 1: skip the multi-rank code path measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="cube edge of the CPU-baseline sample (0 = 1024 if RAM allows, else 512)")
    ap.add_argument("--tune-placement", type=int, default=-1,
                    help="physical backings tried per buffer (work area, out, back) before the warm-up: dfft_tune_placement keeps the one "
                         "the plan's own passes run fastest on.  Default off: buffers come from the library's default backing, which probes its own "
                         "candidates at allocation (dfft_malloc(DFFT_CHUNK_DEFAULT)); --plain-buffers for the caller's plain allocator")
    ap.add_argument("--plain-buffers", action="store_true",
                    help="the reference's ownership contract as it stands: out / back from the caller's plain allocator (torch / hipMalloc) "
                         "and a hipMalloc work area, no tuner (the N = 1 line carries this figure anyway: config.plain_buffers)")
    ap.add_argument("--no-plain-leg", action="store_true", help="N = 1: skip the plain-buffer measurement next to the headline")
    ap.add_argument("--relay", type=int, default=-1,
                    help="N > 1, pencil grids: two-hop relay of the group exchanges (dfft_comm_set_option 'relay'; 1 = exchange 2, 3 = both). "
                         "-1 = both exchanges relayed (3) in a second run of the same plan; the faster of the two runs is the headline, the other "
                         "is config.direct / config.relay; 0 = no relay leg")
    ap.add_argument("--prefer-relay", action="store_true", help="make the relayed run the headline even where it is not faster (tests)")
    ap.add_argument("--no-tune-variants", action="store_true",
                    help="skip dfft_tune_variants (the y / x passes try the streaming sibling of their kernel configuration on the run's own "
                         "buffers before the warm-up; already part of the placement tuner where that runs)")
    ap.add_argument("--pmc", type=int, default=0,
                    help="1 (N = 1, the default workload): measure roofline.traffic in this run -- two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, "
                         "kernel trace only) over tools/kbench on the same grid, after the timed region -- instead of carrying the committed figure")
    ap.add_argument("--dry-run", action="store_true", help="plan C4 / C5 for all 8 ranks without a GPU and print the memory budget")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check (no GPU needed): the ranks meet over gloo, rank 0 prints the world it saw, everybody leaves")
    return ap.parse_args()


def pencil_partition(n):
    # BASELINE.json configs: 1 GPU local passes; 2 GPUs slab; 8 GPUs 2x4 pencil; 4 GPUs 2x2 pencil
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}.get(n, (n, 1))


def choose_partition(n, decomp):
    """The headline is the decomposition BASELINE.json names (pencil 2x4 at 8 GPUs, 2x2 at 4, slab at 2).  On one xGMI
    node every GPU pair has a private link, so a P-way all-to-all drives P-1 links at once: slab (one exchange over all P
    ranks) moves (P-1)/P of the volume once over P-1 links, pencil 2x4 moves 3/4 over 3 links and then 1/2 over a single
    link -- slab is therefore measured next to it as config.alt."""
    if decomp == "slab":
        return (n, 1)
    return pencil_partition(n)


def xgmi_model(esz, N, ngpus, P1, P2):
    """Per transform (one direction) and GPU: exchange 1 runs inside the row group (P2 ranks), exchange 2 inside the column
    group (P1 ranks); every peer pair has its own xGMI link, so an exchange over P ranks sends V/P bytes down each of its
    P - 1 links at the same time (V = bytes of the local volume).  predicted_ms = bytes per link / 153 GB/s."""
    vol = esz * float(N) ** 3 / ngpus
    out = {}
    for name, P in (("exchange 1", P2), ("exchange 2", P1)):
        if P > 1:
            per_link = vol / P
            out[name] = {"group_ranks": P, "links": P - 1, "bytes_per_link": per_link, "bytes_out": per_link * (P - 1),
                         "predicted_ms": round(per_link / (XGMI_LINK_GBS * 1e9) * 1e3, 3)}
            if P < ngpus:
                # two-hop relay (dfft_comm_set_option "relay", csrc/comm.hip): every message is cut into n_gpus parts; all P - 1 partners
                # travel together, so an exchange is TWO grouped operations (hops) in each of which every link of the GPU carries
                # one part of each of the P - 1 messages
                part = per_link / ngpus
                out[name]["relay"] = {"links": ngpus - 1, "phases": 2, "pieces_per_link_per_phase": P - 1,
                                      "bytes_per_link_per_phase": (P - 1) * part,
                                      "predicted_ms": round(2 * (P - 1) * part / (XGMI_LINK_GBS * 1e9) * 1e3, 3)}
    return out


def flops_per_direction(n):
    return 5.0 * float(n) ** 3 * math.log2(float(n) ** 3)


def cpu_baseline(n_req):
    """Times the CPU oracle (oracle/dfft_oracle.c, OpenMP over lines, all host cores) on an n^3 fp64
    complex grid, forward and inverse separately, plus a numpy.fft.fftn cross-check line
    (BASELINE.md 3).  kind = 'port': the reference has no CPU implementation to build."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as orc
    L = orc.lib()
    usable = orc.usable_cores()
    if usable < orc.num_threads():
        orc.set_num_threads(usable)      # one thread per CPU the cgroup grants, not per core of the host
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:   # noqa: BLE001
        avail = 0
    n = n_req or (1024 if avail > 48 * 2 ** 30 else 512)
    g = orc.fill_block((n, n, n), (0, 0, 0), (n, n, n), 2, seed=20260921)
    ptr = g.ctypes.data_as(C.c_void_p)
    L.orc_fft3d_c2c(ptr, n, n, n, -1)      # warm-up (thread pool, page faults)
    L.orc_fft3d_c2c(ptr, n, n, n, +1)
    g /= float(n) ** 3
    tf = tb = 0.0
    iters, t_start = 0, time.perf_counter()
    while True:
        t0 = time.perf_counter()
        L.orc_fft3d_c2c(ptr, n, n, n, -1)
        t1 = time.perf_counter()
        L.orc_fft3d_c2c(ptr, n, n, n, +1)
        t2 = time.perf_counter()
        g /= float(n) ** 3
        tf += t1 - t0
        tb += t2 - t1
        iters += 1
        if time.perf_counter() - t_start > 12.0 or iters >= 3:      # 1 warm-up + up to 3 timed
            break
    fl = flops_per_direction(n)
    # cross-check against an independent implementation (pocketfft) on a 256^3 sample of the same generator
    m = 256
    s = orc.fill_block((m, m, m), (0, 0, 0), (m, m, m), 2, seed=20260921)
    t0 = time.perf_counter()
    want = np.fft.fftn(s)
    t_np = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = orc.fft3d_c2c(s, -1)
    t_or = time.perf_counter() - t0
    dev = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    mpi = cpu_baseline_mpi(n)
    omp = {"value": round(2 * fl * iters / (tf + tb) / 1e9, 3), "cores": orc.num_threads(),
           "forward_ms": round(tf / iters * 1e3, 1), "inverse_ms": round(tb / iters * 1e3, 1),
           "what": "oracle/dfft_oracle.c orc_fft3d_c2c: one process, OpenMP over lines, no decomposition"}
    if mpi.get("value"):
        # Two CPU forms of the same restated path ran: one MPI process per rank (what the reference's CPU/MPI run would be: SURVEY.md
        # 8d) and the single-process OpenMP transform.  `value` is the FASTER of the two -- a baseline must not be picked
        # weak --, `which` says which one it is; both stay in the object.
        best_is_mpi = mpi["value"] >= omp["value"]
        top = mpi if best_is_mpi else omp
        return {"value": top["value"], "unit": "GFLOP/s", "cores": top["cores"], "kind": "port",
                "which": "mpi (one process per rank)" if best_is_mpi else "openmp_single_process",
                "forward_ms": top["forward_ms"], "inverse_ms": top["inverse_ms"],
                "forward_GFLOPs": round(fl / (top["forward_ms"] * 1e-3) / 1e9, 2), "inverse_GFLOPs": round(fl / (top["inverse_ms"] * 1e-3) / 1e9, 2),
                "mpi": mpi, "openmp_single_process": omp,
                "numpy_check": {"grid": f"{m}^3", "numpy_fftn_ms": round(t_np * 1e3, 1), "oracle_ms": round(t_or * 1e3, 1), "max_rel_dev": dev},
                "sample": f"{n}^3 fp64 complex, 1 warm-up + 3 timed forward and inverse transforms each: the restated pencil path as {mpi['cores']} MPI ranks "
                          f"({mpi['P1']}x{mpi['P2']}; mpiexec -n {mpi['cores']} oracle/mpi_pencil: z-FFT, MPI_Alltoallv in the row communicator, y-FFT, "
                          f"MPI_Alltoallv in the column communicator, x-FFT and the mirror: {mpi['value']} GFLOP/s) and the single-process OpenMP oracle on "
                          f"{orc.num_threads()} threads ({omp['value']} GFLOP/s); value = the faster; host has {os.cpu_count()} cores, {mpi['usable_cores']} usable by this job"}
    return {"value": round(2 * fl * iters / (tf + tb) / 1e9, 3), "unit": "GFLOP/s", "cores": orc.num_threads(),
            "kind": "port", "mpi": mpi,
            "forward_ms": round(tf / iters * 1e3, 1), "inverse_ms": round(tb / iters * 1e3, 1),
            "forward_GFLOPs": round(fl * iters / tf / 1e9, 2), "inverse_GFLOPs": round(fl * iters / tb / 1e9, 2),
            "numpy_check": {"grid": f"{m}^3", "numpy_fftn_ms": round(t_np * 1e3, 1), "oracle_ms": round(t_or * 1e3, 1),
                            "max_rel_dev": dev},
            "sample": f"{n}^3 fp64 complex, 1 warm-up + {iters} timed forward and inverse transforms ({tf + tb:.1f} s), "
                      f"oracle/dfft_oracle.c with {orc.num_threads()} OpenMP threads, host has {os.cpu_count()} cores of which the job's cgroup grants {usable}"}


def cpu_baseline_mpi(n):
    """The restated reference path as ONE MPI PROCESS PER RANK on this host's cores (oracle/mpi_pencil.c: the opt1 pencil chain with
    MPI_Comm_split + MPI_Alltoallv, src/pencil/mpicufft_pencil_opt1.cpp:103-104, 1422-1600): R = the largest power of two <= cores
    (at most 256; "cores" = what the cgroup grants), P1 x P2 as square as possible, 1 warm-up + 3 timed forward and inverse transforms of
    the n^3 fp64 complex grid."""
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "oracle", "mpi_pencil")
    launcher = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(exe) or not os.path.exists(launcher):
        return {"error": "oracle/mpi_pencil or mpiexec not available (make -C oracle mpi_pencil)"}
    # the CPUs this job may use: affinity capped by the cgroup quota (the GPU boxes: 256 cores, cpu.max = 16 CPUs; busy-polling MPI
    # ranks beyond the quota only slow each other down -- 1024^3 forward: 11.1 s on 64 ranks, 16.0 s on 128, profiles/r4_mpi_probe.txt)
    from oracle import oracle as orc
    cores = orc.usable_cores()
    R = 1
    while R * 2 <= min(cores, 256):
        R *= 2
    P1 = 1
    while P1 * P1 * 4 <= R:
        P1 *= 2
    P2 = R // P1
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        # 1 warm-up + 3 timed forward and inverse transforms (grids of 512^3 and more)
        out = subprocess.run([launcher, "-n", str(R), exe, str(n), str(P1), str(P2), "3"], capture_output=True, text=True, timeout=400, env=env)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": (out.stderr or out.stdout)[-300:], "ranks": R}
        r = json.loads(line[-1])
    except Exception as e:   # noqa: BLE001
        return {"error": str(e), "ranks": R}
    fl = flops_per_direction(n)
    return {"value": round(2 * fl / ((r["forward_ms"] + r["inverse_ms"]) * 1e-3) / 1e9, 3), "unit": "GFLOP/s", "cores": R, "P1": P1, "P2": P2,
            "forward_ms": round(r["forward_ms"], 1), "inverse_ms": round(r["inverse_ms"], 1), "iters": r["iters"],
            "round_trip_rel_linf": r["round_trip_rel_linf"], "wall_s": round(time.perf_counter() - t0, 1), "usable_cores": cores}


def pmc_traffic_live(N, prec):
    """HBM bytes per launch of the axis-pass kernel measured NOW: tools/pmc_traffic.sh runs tools/kbench (same grid, same library, three
    local passes per transform) under rocprofv3 twice -- FETCH_SIZE and WRITE_SIZE in separate passes, kernel trace only, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes -- and tools/pmc_traffic.py applies the guide's gfx950 corrections."""
    import shutil
    import subprocess
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 is not on PATH"}
    kb = os.path.join(ROOT, "tools", "kbench")
    if not os.path.exists(kb):
        return {"error": "tools/kbench is not built (make -C tools)"}
    tag = "bench_live"
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    esz = 16 if prec == "double" else 8
    try:
        subprocess.run(["bash", os.path.join(ROOT, "tools", "pmc_traffic.sh"), tag, "--", kb, "--size", str(N), "--prec", "f64" if prec == "double" else "f32",
                        "--iters", "2", "--lib-buffers"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(ROOT, "gpurun_out", "pmct_" + tag),
                              str(int(2 * esz * N ** 3)), f"{N}^3 {prec} complex, one axis pass per launch (tools/kbench, measured by bench.py --pmc 1)"],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
        j = json.loads(out.stdout)
        return {k: j.get(k) for k in ("hbm_bytes_per_launch", "read_bytes_per_launch", "write_bytes_per_launch", "algorithmic_bytes_per_launch",
                                      "dispatches", "library_sha256")}
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[-300:]}
    finally:
        shutil.rmtree(os.path.join(ROOT, "gpurun_out", "pmct_" + tag), ignore_errors=True)


def dry_run():
    """Plans BASELINE's C4 (1024^3 fp64, 2x4) and C5 (2048^3 fp32, 2x4) for every rank on this host (no GPU
    needed: decomposition tables only) and prints the per-GPU memory budget."""
    import distributedfft_amd as dfft
    out = {}
    for name, n, prec, (P1, P2) in (("C3", 512, "double", (2, 1)), ("C4", 1024, "double", (2, 4)), ("C5", 2048, "float", (2, 4)),
                                    ("C4-slab", 1024, "double", (8, 1)), ("C5-slab", 2048, "float", (8, 1))):
        esz = 16 if prec == "double" else 8
        world = dfft.Comm.local(P1 * P2)
        per_rank = []
        for r in range(P1 * P2):
            pl = (dfft.MPIcuFFT_Slab_Opt1 if P2 == 1 else dfft.MPIcuFFT_Pencil_Opt1)(dfft.Configurations(), world, precision=prec, rank=r)
            pl.initFFT(dfft.GlobalSize(n, n, n), dfft.Partition(P1, P2), allocate=False, c2c=True)
            isz = pl.getInSize()
            in_b = isz[0] * isz[1] * isz[2] * esz
            per_rank.append({"in": in_b, "out": pl.getDomainSize(), "work": pl.getWorkSizeDevice(),
                             "total": 2 * in_b + pl.getDomainSize() + pl.getWorkSizeDevice()})
        worst = max(per_rank, key=lambda d: d["total"])
        out[name] = {"grid": f"{n}^3 {prec}", "partition": f"{P1}x{P2}", "domain_GiB": round(worst["out"] / 2 ** 30, 3),
                     "work_GiB": round(worst["work"] / 2 ** 30, 3),
                     "per_gpu_total_GiB (in + out + back + work)": round(worst["total"] / 2 ** 30, 3),
                     "fits_288_GB": worst["total"] < 268 * 2 ** 30}
    print(json.dumps({"dry_run": out}))


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves, one process per GPU on this node
    (the reference's launcher builds its own `mpiexec -n P ...` line the same way, launch.py:168-247).  The child processes are
    this script under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`
    with the caller's argv; rank 0's JSON line goes to our stdout, the return code is non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["DFFT_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd[1:8])} bench.py ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.dry_run:
        return dry_run()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ngpus = args.gpus
    if world != ngpus:
        if rank == 0:
            print(f"[bench] --gpus {ngpus} but WORLD_SIZE = {world}: running on {world} ranks", file=sys.stderr, flush=True)
        ngpus = world
    dist = None
    if args.rendezvous_only:
        # launcher check that needs no GPU (tests/test_cpu_multiprocess.py): the ranks meet over gloo, agree on the world and leave
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t)
            total = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            total = 1
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world_size": world, "sum_of_ranks_plus_1": total,
                              "self_launched": os.environ.get("DFFT_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        return None

    import distributedfft_amd as dfft

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    ndev = torch.cuda.device_count()
    if world > ndev and args.backend == "nccl":
        sys.exit(f"bench.py --gpus {world}: {ndev} device(s) visible; RCCL needs one GPU per rank (--backend gloo lets ranks share a GPU, functional test only)")
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    if world > ndev:
        # several ranks per GPU (the gloo functional runs): each keeps its allocator's search within its share of the device
        os.environ.setdefault("DFFT_RANKS_PER_DEVICE", str(-(-world // ndev)))
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    N = args.size
    prec = args.precision
    esz = 16 if prec == "double" else 8
    cdt = torch.complex128 if prec == "double" else torch.complex64
    rdt = torch.float64 if prec == "double" else torch.float32
    P1, P2 = (args.p1, args.p2) if args.p1 and args.p2 else choose_partition(ngpus, args.decomp)
    assert P1 * P2 == ngpus

    # a dedicated (non-default) torch stream carries the plan's kernels AND, being torch's current
    # stream inside `with torch.cuda.stream(side)`, the collectives of the torch transport
    side = torch.cuda.Stream()
    stream = side.cuda_stream
    tmode = "torch" if args.backend == "gloo" else args.transport
    tmode_box = [tmode]

    def lib_buffer(nbytes, dtype):
        """device memory with the library's default backing (dfft_malloc(DFFT_CHUNK_DEFAULT): virtual-memory API, no search) as a
        torch view -- what a drop-in caller gets by allocating `out` through the library; --plain-buffers: the caller's allocator"""
        if args.plain_buffers:
            return torch.empty(nbytes // torch.empty((), dtype=dtype).element_size(), dtype=dtype, device="cuda")
        return dfft.DeviceBuffer.alloc(nbytes).tensor(dtype)

    def make_plan(P1, P2, options=None):
        comm, transport = None, "none"
        if world > 1:
            from distributedfft_amd.torch_transport import make_comm
            comm, transport = make_comm(dist, rank, world, P1, P2, tmode_box[0])
            if transport.startswith("rccl") and P1 > 1 and P2 > 1 and not args.no_dup_channel:
                # row- and column-group exchanges use disjoint links: give the second one its own communicator (collective)
                # (3: one for exchange 2 and one per exchange for the relay's first hop, which runs under the second hop of the chunk before)
                try:
                    comm.setOption("dup_channel", 3)
                    transport += ", duplicated communicators for exchange 2 and the relay's first hops"
                except Exception as e:   # noqa: BLE001
                    if rank == 0:
                        print(f"[bench] dup_channel unavailable: {e}", file=sys.stderr, flush=True)
        kind = dfft.MPIcuFFT_Slab_Opt1 if P2 == 1 and ngpus > 1 else dfft.MPIcuFFT_Pencil_Opt1
        plan = kind(dfft.Configurations(), comm, precision=prec, rank=rank)
        # The chunks of a pass on ONE compute stream: the timed region prices the axis-pass kernel by its own launches (roofline:
        # algorithmic bytes / average launch duration, from events around each launch), which only means something for launches that
        # own the device; the library's default from three chunks on is two streams (chunk c + 1 ramps under the drain of chunk c:
        # 2-3 % of the per-GPU kernel time, config.per_gpu_kernels_8gpu reports it as step_ms against step_ms_one_stream).
        plan.setOption("compute_streams", 1)
        for k, v in (options or {}).items():
            plan.setOption(k, v)
        plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(P1, P2), allocate=False, c2c=True)
        plan.setStream(stream)
        if comm is not None and transport == "torch":
            # the torch transport maps raw pointers back to tensors, so the work area must be one (a view of library memory)
            work = lib_buffer(plan.getWorkSizeDevice(), torch.uint8)
            plan.setWorkArea(work)
            comm.register(work)
        elif args.plain_buffers:
            work = torch.empty(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
            plan.setWorkArea(work)
        else:
            work = None
            plan.setWorkArea(None)       # library-owned (the default backing), like the reference's allocate = true
        return plan, comm, transport, work

    plan, comm, transport, work = make_plan(P1, P2)
    domain = plan.getDomainSize()

    # synthetic input: this rank's block of a uniform[0,255) complex grid (the reference scales
    # cuRAND uniforms by 255, tests/src/pencil/base.cu:45-53), generated on the device
    isz = plan.getInSize()
    n_in = isz[0] * isz[1] * isz[2]
    chunk = 1 << 26

    def fill(t):
        gen = torch.Generator(device="cuda")
        gen.manual_seed(20260921 + rank)
        v = torch.view_as_real(t)
        for o in range(0, n_in, chunk):
            e = min(n_in, o + chunk)
            v[o:e] = torch.rand((e - o, 2), dtype=rdt, device="cuda", generator=gen) * 255.0

    def round_trip_error(back):
        """max |back/N^3 - x| / max |x| against the regenerated input (x itself may have been overwritten)"""
        gen = torch.Generator(device="cuda")
        gen.manual_seed(20260921 + rank)
        worst = torch.zeros((), dtype=torch.float64, device="cuda")
        v = torch.view_as_real(back)
        for o in range(0, n_in, chunk):
            e = min(n_in, o + chunk)
            ref = torch.rand((e - o, 2), dtype=rdt, device="cuda", generator=gen) * 255.0
            worst = torch.maximum(worst, (v[o:e] / float(N) ** 3 - ref).abs().max().to(torch.float64))
        e = (worst / 255.0).reshape(1)
        if dist is not None:
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
        return float(e)

    d_in = torch.empty(n_in, dtype=cdt, device="cuda")
    fill(d_in)
    free_b, _ = torch.cuda.mem_get_info()
    aliased = free_b < domain + n_in * esz + (2 << 30)     # 2048^3 fp32 on one GPU: the inverse writes back over the input
    # Placement: the passes that scatter 128-byte runs depend on the physical backing of the buffer they write
    # to.  Before the warm-up the plan tries a few backings for its work area and for out / back (virtual-memory API, chunks
    # of different sizes) and keeps the fastest; that needs room for two candidates of a buffer at a time.
    # (round 4: the library's default backing probes its own candidates -- local, 0-5 s, csrc/dfft.hip dev_alloc_default -- and lands
    # in the same range as the plan-level search, so that search is no longer the default; --tune-placement 6 brings it back)
    tries = args.tune_placement if args.tune_placement >= 0 else 0
    if args.plain_buffers:
        tries = 0
    placement = None
    if tries > 1 and work is None and not aliased and free_b > 4 * domain + (4 << 30):      # (fewer candidates where memory is short: the library checks before each)
        t_tune = time.perf_counter()
        b_out, b_back, trial_ms = plan.tunePlacement(d_in, tries, want_back=True)
        d_out = b_out.tensor(cdt)
        d_back = b_back.tensor(cdt)[:n_in]
        placement = {"tries_per_buffer": tries, "trial_fft_ms_fwd_plus_inv": [round(v, 3) for v in trial_ms],
                     "what": "dfft_tune_placement before the warm-up: first entry = plain hipMalloc everywhere, then the work area, out and "
                             "back one at a time on other physical backings; a candidate is kept when the plan's own passes run faster on it",
                     "seconds": round(time.perf_counter() - t_tune, 2)}
    else:
        t_alloc = time.perf_counter()
        d_out = lib_buffer(domain, cdt)
        info_out = None if args.plain_buffers else dfft.last_placement_info()
        d_back = d_in if aliased else lib_buffer(n_in * esz, cdt)
        info_back = None if (args.plain_buffers or aliased) else dfft.last_placement_info()
        t_alloc = time.perf_counter() - t_alloc
        placement = {"tries_per_buffer": 0, "alloc_seconds_out_and_back": round(t_alloc, 2), "what": "no search: out / back from the caller's plain allocator, hipMalloc work area (--plain-buffers)"
                     if args.plain_buffers else "no plan-level search: out / back from dfft_malloc(DFFT_CHUNK_DEFAULT) and the library-owned work area on the "
                     "same default backing (virtual-memory API, 1 GiB physical chunks; buffers of 1 GiB and more: "
                     "built from chunks K apart, probed with a streaming write against the device's contiguous reference, else up to 6 drawn "
                     "candidates, the fastest kept -- local to the device, bounded by half of the free memory, csrc/dfft.hip dev_alloc_default)"}
        if info_out and info_out.get("bytes") == domain:
            placement["out"] = info_out
        if info_back and info_back.get("bytes") == n_in * esz:
            placement["back"] = info_back
    if comm is not None and transport == "torch":
        comm.register(d_out)
    torch.cuda.synchronize()
    variants = None
    if not (placement or {}).get("tries_per_buffer") and not args.no_tune_variants:
        # no placement tuner in this run (N > 1, or a grid that leaves no room for candidates): the kernel-configuration half of it
        # on the buffers as they are.  Collective at N > 1 (it executes the plan); every rank decides for its own kernels.
        try:
            with torch.cuda.stream(side):
                trial_ms = plan.tuneVariants(d_in, d_out, None if aliased else d_back)
            variants = {"trial_fft_ms": [round(v, 3) for v in trial_ms], "chosen (variant, order, addr64) per pass": plan.getPassChoices(),
                        "what": "dfft_tune_variants before the warm-up: first entry = the plan as built (rule-based workgroup orders and kernel "
                                "configurations), then the four workgroup-order settings (each pass keeps its fastest), the chosen orders, one entry per "
                                "kernel-configuration number of the plan's line lengths (each pass keeps a configuration that is more than 1 % faster), "
                                "the final choice"}
        except Exception as e:   # noqa: BLE001
            variants = {"error": str(e)}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(pl, k, out_buf, back_buf, collect=False):
        """k forward + inverse pairs; returns (seconds, {phase name: total ms}, launches of FFT passes).
        Normally ONE bracket (barrier + synchronize on both sides) around all k steps.  When the inverse writes back over
        the input (`aliased`: 2048^3 fp32 on one GPU) every step multiplies the data by N^3, which overflows fp32 within
        four steps: each step then gets its own bracket and the input is regenerated between the brackets, outside the
        timed spans; the reported time is the sum of the k brackets."""
        acc, launches = {}, 0
        refill = back_buf is d_in
        dt = 0.0
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            if refill and i:
                fill(d_in)
                barrier()
                t0 = time.perf_counter()
            with torch.cuda.stream(side):
                pl.execC2C(out_buf, d_in, dfft.FORWARD)       # blocking, like the reference's exec
                ph_f = pl.getPhaseTimes(dfft.FORWARD) if collect else []
                pl.execC2C(back_buf, out_buf, dfft.INVERSE)
                ph_b = pl.getPhaseTimes(dfft.INVERSE) if collect else []
            for name, ms in ph_f + ph_b:
                acc[name] = acc.get(name, 0.0) + ms
                launches += "FFT" in name
            if refill:
                barrier()
                dt += time.perf_counter() - t0
        if not refill:
            barrier()
            dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            if acc:
                names = sorted(acc)
                t = torch.tensor([acc[n] for n in names], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)      # max over ranks of the per-phase device times
                acc = dict(zip(names, t.tolist()))
        return dt, acc, launches

    def warm_and_check(pl, out_buf, back_buf):
        run_steps(pl, args.warmup, out_buf, back_buf)
        if args.warmup == 0:
            return None
        return round_trip_error(back_buf)      # round-trip check on the warm-up result (reference testcase 3), worst rank

    rt_err = warm_and_check(plan, d_out, d_back)
    assert rt_err is None or math.isfinite(rt_err), f"round trip is not finite ({rt_err})"
    tol = 1e-10 if prec == "double" else 5e-5
    if rt_err is not None and not rt_err < tol and world > 1 and transport.startswith("rccl") and args.transport == "auto":
        # the native RCCL exchange produced a wrong round trip at full size: measure with the torch
        # transport instead (still RCCL underneath) and say so in `config.transport`
        if rank == 0:
            print(f"[bench] native RCCL transport failed the full-size round trip ({rt_err}); using torch transport",
                  file=sys.stderr, flush=True)
        tmode_box[0] = "torch"
        plan, comm, transport, work = make_plan(P1, P2)
        comm.register(d_out)
        transport = "torch (fallback: native RCCL failed the full-size round trip)"
        if aliased:
            fill(d_in)
        rt_err = warm_and_check(plan, d_out, d_back)
    if aliased:
        fill(d_in)          # the timed region must start from the input, not from a round-tripped copy

    plan.enablePhaseTiming(True)
    dt, phases, kern_launches = run_steps(plan, args.steps, d_out, d_back, collect=True)
    kern_ms = sum(ms for name, ms in phases.items() if "FFT" in name)
    exch_ms = sum(ms for name, ms in phases.items() if "FFT" not in name)

    vol_bytes = 2.0 * esz * float(N) ** 3 / ngpus      # one axis pass reads the local volume once and writes it once

    def per_pass(ph, steps):
        out = {}
        for name, ms in ph.items():
            m = ms / steps
            out[name] = {"ms": round(m, 3)}
            if "FFT" in name and m > 0:
                out[name]["TBps"] = round(vol_bytes / (m * 1e-3) / 1e12, 3)
        return out

    def overlap_report(step_ms, fft_ms, exch_ms):
        """hidden_frac = 1 - (step - sum of kernel spans) / (sum of exchange spans): 1 = every exchange span ran under a
        kernel, 0 = none did (the spans are device times of the exchange calls on the communication streams)"""
        exposed = max(step_ms - fft_ms, 0.0)
        return {"step_ms": round(step_ms, 3), "kernels_ms": round(fft_ms, 3), "exchanges_ms": round(exch_ms, 3),
                "exposed_ms": round(exposed, 3),
                "hidden_frac": round(1.0 - exposed / exch_ms, 3) if exch_ms > 0 else None}

    def per_gpu_kernels(P1v, P2v, steps, options=None, real=False):
        """rank 0's plan of a P1v x P2v grid on THIS GPU with the exchange stubbed out (a callback transport that moves
        nothing): the kernels run with that rank's descriptors -- 1/P of the volume, its peer segments, its pipeline chunks --
        on whatever the buffers hold, so the times are the compute one GPU of the multi-GPU run does per step.
        Two plans: one with the chunks of a pass serialised on ONE compute stream (option compute_streams = 1), whose per-pass
        device spans can be summed (`per_pass`, `kernels_ms_per_step`), and the plan as a rank would run it (two compute streams
        from three chunks on: chunk spans overlap, so only the whole step is timed: `step_ms`, next to `step_ms_one_stream`)."""
        nr = P1v * P2v
        kind = dfft.MPIcuFFT_Slab_Opt1 if P2v == 1 else dfft.MPIcuFFT_Pencil_Opt1

        def build(extra):
            stub = dfft.Comm.callback(nr, 0, lambda *a: None)
            pl = kind(dfft.Configurations(), stub, precision=prec, rank=0)
            for k, v in {**(options or {}), **extra}.items():
                pl.setOption(k, v)
            pl.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(P1v, P2v), allocate=False, c2c=not real)
            pl.setStream(stream)
            pl.setWorkArea(None)
            isz_v = pl.getInSize()
            nv = isz_v[0] * isz_v[1] * isz_v[2]
            # real = the reference's own API: execR2C / execC2R (real input block [xs][ys][Nz], Hermitian half [Nx][yo][zs] out)
            v_in = torch.view_as_real(d_in).reshape(-1)[:nv] if real else d_in[:nv]
            v_out = d_out[:pl.getDomainSize() // esz]
            v_back = None if aliased else (torch.view_as_real(d_back).reshape(-1)[:nv] if real else d_back[:nv])

            def fwd():
                if real:
                    pl.execR2C(v_out, v_in)
                else:
                    pl.execC2C(v_out, v_in, dfft.FORWARD)

            def inv():
                if real:
                    pl.execC2R(v_in, v_out)
                else:
                    pl.execC2C(v_in, v_out, dfft.INVERSE)
            tuned = None
            if not args.no_tune_variants:
                # the plan's own tuner (dfft_tune_variants: workgroup order and kernel configuration per pass, by measurement) -- what
                # bench.py calls at N > 1 before the warm-up, so these are the kernels a rank of the 8-GPU run would launch
                try:
                    with torch.cuda.stream(side):
                        tr = pl.tuneVariants(v_in, v_out, v_back)   # without a third buffer: forward passes only
                    tuned = {"as_built_ms": round(tr[0], 3), "chosen_ms": round(tr[-1], 3), "trials": len(tr),
                             "chosen (variant, order, addr64) per pass": pl.getPassChoices()}
                except Exception as e:   # noqa: BLE001
                    tuned = {"error": str(e)}
            return stub, pl, fwd, inv, tuned

        def whole_step_ms(fwd, inv):
            """forward + inverse pairs enqueued back to back, no phase events, one synchronisation at the end"""
            with torch.cuda.stream(side):
                fwd(); inv()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(side):
                for _ in range(steps):
                    fwd(); inv()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        stub, pl, fwd, inv, tuned = build({"compute_streams": 1})
        one_stream_ms = whole_step_ms(fwd, inv)
        pl.enablePhaseTiming(True)
        acc = {}
        for i in range(steps + 2):
            with torch.cuda.stream(side):
                fwd()
                ph = pl.getPhaseTimes(dfft.FORWARD)
                inv()
                ph = ph + pl.getPhaseTimes(dfft.INVERSE)
            if i >= 2:
                for name, ms in ph:
                    if "FFT" in name:
                        acc[name] = acc.get(name, 0.0) + ms
        torch.cuda.synchronize()
        osz_v = pl.getOutSize()
        chunks_v = pl.getPipelineChunks()
        # bytes one pass moves: the local volume read once and written once (R2C: the Hermitian half of this rank, [Nx][yo][zs])
        vb = 2.0 * esz * float(osz_v[0] * osz_v[1] * osz_v[2]) if real else 2.0 * esz * float(N) ** 3 / nr
        passes = {name: {"ms": round(ms / steps, 3), "TBps": round(vb / (ms / steps * 1e-3) / 1e12, 3)} for name, ms in acc.items()}
        tot = sum(v["ms"] for v in passes.values())
        del fwd, inv, pl          # (the closures hold the plan: its work area goes with the last reference)
        stub.destroy()
        step_ms, streams = one_stream_ms, 1
        if (options or {}).get("compute_streams", -1) != 1 and chunks_v >= 3:
            stub, pl, fwd, inv, _ = build({})
            step_ms, streams = whole_step_ms(fwd, inv), 2
            del fwd, inv, pl
            stub.destroy()
        res = {"decomposition": f"slab P={P1v}" if P2v == 1 else f"pencil {P1v}x{P2v}", "rank": 0,
               "pipeline_chunks": chunks_v, "compute_streams": streams,
               "step_ms": round(step_ms, 3), "step_ms_one_stream": round(one_stream_ms, 3),
               "step_avg_TBps": round(6 * vb / (step_ms * 1e-3) / 1e12, 3) if step_ms > 0 else None,
               "kernels_ms_per_step": round(tot, 3), "per_pass": passes,
               "alg_bytes_per_pass": vb, "avg_TBps": round(6 * vb / (tot * 1e-3) / 1e12, 3) if tot > 0 else None,
               "tune_variants": tuned, "xgmi_model_per_transform": xgmi_model(esz, N, nr, P1v, P2v)}
        if options:
            res["options"] = dict(options)
        if real:
            res["transform"] = "execR2C + execC2R (the reference's own API); bytes per pass = 2 x the rank's Hermitian half"
        return res

    chunks_main = plan.getPipelineChunks()
    flops_step = 2 * flops_per_direction(N)
    ms_per_step = dt / args.steps * 1e3
    value = flops_step * args.steps / dt / 1e9

    # roofline of the dominant kernel: fft_pass_kernel.  One axis pass (one launch on a single
    # GPU; `pipeline_chunks` launches when the pass is pipelined against an exchange) reads the
    # local volume once and writes it once: algorithmic bytes = 2 * esz * N^3 / n_gpus
    # (SURVEY.md 8d).  Duration = HIP events around the launches on the launch stream.
    avg_ms = kern_ms / max(kern_launches, 1)
    achieved = vol_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy_peak": round(achieved / HBM_COPY_PEAK_GBS, 4), "traffic": None,
                "kernel": "dfft::fft_pass_kernel", "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": kern_launches, "alg_bytes_per_launch": vol_bytes,
                "launches_per_pass": chunks_main if ngpus > 1 else 1}

    # the same launch against the COMPUTE roof (north_star: "FLOP/s for the butterfly pass against the gfx950 roofline"): one axis pass
    # is 5 N^3 log2 N flops by the metric's own convention; FP64 vector peak measured on an MI355X with tools/fp64_peak
    # (profiles/r6_fp64_peak.txt: 73.4 TFLOP/s; public specification 78.6 = 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz; the in-image
    # guide lists no FP64 figure).  The pass sits at ~13 % of it: bandwidth-bound by a factor of six, as the intensity says
    # (5 log2 N / 32 B = 1.56 flop/B at N = 1024 fp64 -> 9.8 TFLOP/s at the 6.29 TB/s copy rate).
    flops_launch = 5.0 * float(N) ** 3 * math.log2(N) / ngpus
    peak_tf = FP64_VECTOR_PEAK_TFLOPS if prec == "double" else FP32_VECTOR_PEAK_TFLOPS
    ach_tf = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    roofline["compute"] = {"flops_per_launch": flops_launch, "achieved_TFLOPs": round(ach_tf, 2), "peak_TFLOPs": peak_tf,
                           "frac": round(ach_tf / peak_tf, 4),
                           "peak_source": ("measured: tools/fp64_peak, profiles/r6_fp64_peak.txt (specification 78.6)" if prec == "double"
                                           else "specification: FP32 vector 157.3 TFLOP/s (packed / dual-issue FMA)"),
                           "flop_per_byte": round(flops_launch / vol_bytes, 3)}

    # HBM bytes per launch: NOT measured by this run.  It is the figure of the committed PMC profile of this kernel
    # on this workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read correction applied)
    # The committed figure is only carried if it was measured on THIS library: the profile records the sha256 of the libdfft_amd.so
    # it ran; a different library loaded here means the figure is stale (traffic stays null, traffic_stale says why).
    try:
        import hashlib
        so = os.path.join(ROOT, "distributedfft_amd", "libdfft_amd.so")
        sha = hashlib.sha256(open(so, "rb").read()).hexdigest()
        roofline["library_sha256"] = sha
        # newest committed PMC run of the headline kernel (tools/pmc_traffic.sh + tools/pmc_traffic.py)
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
        loaded = [(os.path.basename(f), json.load(open(f))) for f in cands]
        match = [x for x in loaded if x[1].get("library_sha256") == sha]
        pmc, pm = (match or loaded)[0]      # the profile of THIS library if there is one, else the newest (reported as stale)
        if ngpus == 1 and N == 1024 and prec == "double" and pm.get("hbm_bytes_per_launch"):
            roofline["traffic_source"] = f"profiles/{pmc} (a committed rocprofv3 PMC run, not this run)"
            if pm.get("library_sha256") == sha:
                roofline["traffic"] = pm["hbm_bytes_per_launch"]
                roofline["traffic_static"] = True
            else:
                roofline["traffic_stale"] = True
                roofline["traffic_stale_value"] = pm["hbm_bytes_per_launch"]
                roofline["traffic_stale_why"] = ("the profile was measured on library sha256 %s, this run loaded %s" %
                                                 (str(pm.get("library_sha256"))[:12], sha[:12]))
    except Exception:   # noqa: BLE001
        pass

    if args.pmc and ngpus == 1:
        live = pmc_traffic_live(N, prec)
        roofline["traffic_live"] = live
        if live.get("hbm_bytes_per_launch"):
            roofline["traffic"] = live["hbm_bytes_per_launch"]
            roofline["traffic_static"] = False
            roofline["traffic_source"] = "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/kbench on the same grid and library"
            for k in ("traffic_stale", "traffic_stale_value", "traffic_stale_why"):
                roofline.pop(k, None)

    rccl_nranks = comm.info()[1] if (comm is not None and hasattr(comm, "info")) else 0
    out = None
    if rank == 0:
        out = {
            "metric": "3D FFT GFLOP/s (5N^3 log2 N^3 per direction), forward+inverse",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": ngpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64" if prec == "double" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{N}^3 {'fp64' if prec == 'double' else 'fp32'} complex forward+inverse",
                       "decomposition": "single GPU, three local axis passes" if ngpus == 1 else
                       (f"slab P={P1}" if P2 == 1 else f"pencil {P1}x{P2}"),
                       "transport": transport, "exchange_ms_per_step": round(exch_ms / args.steps, 3),
                       "fft_ms_per_step": round(kern_ms / args.steps, 3),
                       "pipeline_chunks": chunks_main,
                       # what the transport itself says (ncclCommCount; 0 = the torch transport, ask torch.distributed)
                       "rccl_nranks": rccl_nranks, "world_size": world, "devices_visible": ndev,
                       "ranks_per_device": max(1, -(-world // ndev)),
                       "per_pass": per_pass(phases, args.steps),
                       "input_aliased_with_inverse_output": bool(aliased), "placement": placement, "variants": variants},
            "round_trip_rel_linf": rt_err,
            "roofline": roofline,
        }

    # N = 1: the same plan on the buffers a caller gets from its own allocator (the reference's ownership contract as it stands:
    # cudaMalloc'd in / out, tests/src/pencil/random_dist_3D.cu:197-205) and a hipMalloc work area, no tuner of any kind
    plain_leg = None
    free_b, _ = torch.cuda.mem_get_info()
    if ngpus == 1 and not args.no_plain_leg and not args.plain_buffers and not aliased and free_b > 3 * domain + n_in * esz + (4 << 30):
        pp = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), None, precision=prec, rank=0)
        pp.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), allocate=False, c2c=True)
        pp.setStream(stream)
        p_work = torch.empty(pp.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
        pp.setWorkArea(p_work)
        p_out = torch.empty(domain // esz, dtype=cdt, device="cuda")
        p_back = torch.empty(n_in, dtype=cdt, device="cuda")
        ksteps = max(1, min(args.steps, 10))
        run_steps(pp, 2, p_out, p_back)
        rt_p = round_trip_error(p_back)
        pp.enablePhaseTiming(True)
        dtp, php, _ = run_steps(pp, ksteps, p_out, p_back, collect=True)
        plain_leg = {"what": "the same plan on plain buffers: out / back from torch's allocator (hipMalloc), hipMalloc work area, rule-based "
                             "kernel configurations, no tuner -- what a caller that follows the reference's ownership contract to the letter gets",
                     "ms_per_step": round(dtp / ksteps * 1e3, 3), "steps": ksteps, "round_trip_rel_linf": rt_p,
                     "value_GFLOPs": round(2 * flops_per_direction(N) * ksteps / dtp / 1e9, 1), "per_pass": per_pass(php, ksteps)}
        del pp, p_work
        # ... and what an UNMODIFIED reference call site gets: its own cudaMalloc'd in / out / back, and a work
        # area the library owns (initFFT(..., allocate = true), placed by the library's allocator); no tuner call
        unmodified = None
        try:
            pu = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), None, precision=prec, rank=0)
            pu.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), allocate=False, c2c=True)
            pu.setStream(stream)
            pu.setWorkArea(None)
            run_steps(pu, 2, p_out, p_back)
            rt_u = round_trip_error(p_back)
            pu.enablePhaseTiming(True)
            dtu, phu, _ = run_steps(pu, ksteps, p_out, p_back, collect=True)
            unmodified = {"what": "an unmodified reference call site: out / back from the caller's plain allocator (hipMalloc), the work area owned and "
                                  "placed by the library (allocate = true), rule-based kernel configurations, no tuner call",
                          "ms_per_step": round(dtu / ksteps * 1e3, 3), "steps": ksteps, "round_trip_rel_linf": rt_u,
                          "value_GFLOPs": round(2 * flops_per_direction(N) * ksteps / dtu / 1e9, 1), "per_pass": per_pass(phu, ksteps)}
            del pu
        except Exception as e:   # noqa: BLE001
            unmodified = {"error": str(e)}
        plain_leg["unmodified_caller"] = unmodified
        del p_out, p_back
        torch.cuda.empty_cache()

    # N > 1, pencil grids: the same plan with the two-hop relay on its group exchanges (every rank sets the option; collective)
    relay_leg = None
    want_relay = args.relay if args.relay >= 0 else 3      # both exchanges of a pencil grid run inside strict subsets of the world
    if world > 1 and comm is not None and want_relay and (P1 < world and P1 > 1 or (want_relay & 2 and P2 < world and P2 > 1)):
        # The relay has run on virtual ranks and over gloo only (no lease here has two GPUs).  A collective that hangs cannot be caught
        # by an exception handler, so a watchdog guards this leg: if it does not finish in time, rank 0 prints the headline line as
        # measured so far (config.relay says what happened) and every rank leaves; the headline never depends on this leg.
        import threading

        def bail(why=None):
            if rank == 0:
                out["config"]["relay"] = {"error": why or "the relayed run did not finish within %d s: abandoned, the line holds the direct run only" % RELAY_TIMEOUT_S}
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(RELAY_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            comm.setOption("relay", want_relay)
            if aliased:
                fill(d_in)
            run_steps(plan, 2, d_out, d_back)
            rt_r = round_trip_error(d_back)
            if aliased:
                fill(d_in)
            dtr, phr, launches_r = run_steps(plan, args.steps, d_out, d_back, collect=True)
            exr = {name: round(ms / args.steps / 2.0, 3) for name, ms in phr.items() if "FFT" not in name}
            relay_leg = {"what": "the headline plan with dfft_comm_set_option(comm, 'relay', %d): every message of the relayed exchanges cut into "
                                 "n_gpus parts, two direct and the others through the ranks outside the pair; all partners together: two grouped "
                                 "send/receive operations per exchange and pipeline chunk, hop 1 of a chunk under hop 2 of the chunk before (csrc/comm.hip)" % want_relay,
                         "relay": want_relay, "ms_per_step": round(dtr / args.steps * 1e3, 3), "round_trip_rel_linf": rt_r,
                         "value": round(2 * flops_per_direction(N) * args.steps / dtr / 1e9, 1),
                         "exchange_ms_per_transform": exr, "per_pass": per_pass(phr, args.steps),
                         "transport_counters": comm.comm.counters() if hasattr(comm, "comm") else comm.counters()}
            relay_leg["overlap"] = overlap_report(relay_leg["ms_per_step"], sum(ms for n_, ms in phr.items() if "FFT" in n_) / args.steps,
                                                  sum(ms for n_, ms in phr.items() if "FFT" not in n_) / args.steps)
        except Exception as e:   # noqa: BLE001
            # a failure on this rank leaves the others inside a collective: the job cannot go on together.  This rank leaves at once
            # (rank 0 with the direct line), the others when their watchdogs fire.
            bail("the relayed run failed on rank %d: %s; the line holds the direct run only" % (rank, e))
        finally:
            watchdog.cancel()
            comm.setOption("relay", 0)
        # The relay changes how the bytes of an exchange travel, not the decomposition or the work: where the relayed run of the SAME
        # plan, timed by the same protocol, is faster and its round trip is within tolerance, it is the headline and the direct run is
        # kept as config.direct (dtr and dt are maxima over the ranks: every rank takes the same branch)
        if rank == 0:
            # the headline of a pencil grid is the better of two runs of the same plan (exchanges direct / relayed): say so at the top level
            out["headline_is_best_of"] = 2
            out["headline_candidates"] = {"direct_ms_per_step": out["ms_per_step"], "relayed_ms_per_step": relay_leg.get("ms_per_step")}
        if "error" not in relay_leg and relay_leg["round_trip_rel_linf"] < tol and (dtr < dt or args.prefer_relay):
            relay_leg["headline"] = True
            kern_r = sum(ms for n_, ms in phr.items() if "FFT" in n_)
            if rank == 0:
                out["config"]["direct"] = {"what": "the same plan, same protocol, exchanges sent directly (no relay)", "value": out["value"],
                                           "ms_per_step": out["ms_per_step"], "round_trip_rel_linf": out["round_trip_rel_linf"],
                                           "per_pass": out["config"]["per_pass"], "exchange_ms_per_step": out["config"]["exchange_ms_per_step"],
                                           "fft_ms_per_step": out["config"]["fft_ms_per_step"], "roofline_avg_launch_ms": out["roofline"]["avg_launch_ms"]}
                out["value"], out["ms_per_step"], out["round_trip_rel_linf"] = relay_leg["value"], relay_leg["ms_per_step"], relay_leg["round_trip_rel_linf"]
                out["config"]["transport"] = transport + f" + two-hop relay of the group exchanges (relay = {want_relay})"
                out["config"]["per_pass"] = relay_leg["per_pass"]
                out["config"]["fft_ms_per_step"] = round(kern_r / args.steps, 3)
                out["config"]["exchange_ms_per_step"] = round(sum(ms for n_, ms in phr.items() if "FFT" not in n_) / args.steps, 3)
                avg_r = kern_r / max(launches_r, 1)
                ach = vol_bytes / (avg_r * 1e-3) / 1e9 if avg_r > 0 else 0.0
                out["roofline"].update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                                        "frac_of_measured_copy_peak": round(ach / HBM_COPY_PEAK_GBS, 4), "avg_launch_ms": round(avg_r, 4),
                                        "launches_timed": launches_r})
            dt, phases, kern_ms = dtr, phr, kern_r
            exch_ms = sum(ms for n_, ms in phr.items() if "FFT" not in n_)
            ms_per_step = dt / args.steps * 1e3

    # N = 1: the code path of the N > 1 runs on the same grid (mirrored inverse order, 8-chunk segment tables)
    multi_rank_path = None
    free_b, _ = torch.cuda.mem_get_info()
    if ngpus == 1 and not args.no_multi_rank_path and free_b < plan.getWorkSizeDevice() + (2 << 30):
        del plan            # 2048^3 fp32: only one work area fits next to the grid
        plan = None
    if ngpus == 1 and not args.no_multi_rank_path:
        plan_m, _, _, _ = make_plan(1, 1, {"mirror_inverse": 1, "pipeline_chunks": 8})
        ksteps = max(1, min(args.steps, 10))
        tuned_m = None
        if not args.no_tune_variants:
            try:
                with torch.cuda.stream(side):
                    tr = plan_m.tuneVariants(d_in, d_out, None if aliased else d_back)
                tuned_m = {"as_built_ms": round(tr[0], 3), "chosen_ms": round(tr[-1], 3), "trials": len(tr)}
            except Exception as e:   # noqa: BLE001
                tuned_m = {"error": str(e)}
            if aliased:
                fill(d_in)
        run_steps(plan_m, 2, d_out, d_back)
        rt_m = round_trip_error(d_back)
        if aliased:
            fill(d_in)
        plan_m.enablePhaseTiming(True)
        dtm, phm, _ = run_steps(plan_m, ksteps, d_out, d_back, collect=True)
        multi_rank_path = {"what": "same grid, one GPU: inverse in the multi-rank pass order x^-1, y^-1, z^-1 (strided read of the "
                                   "API layout), every pass cut into 8 pipeline chunks with segmented address tables",
                           "ms_per_step": round(dtm / ksteps * 1e3, 3), "steps": ksteps, "round_trip_rel_linf": rt_m,
                           "value_GFLOPs": round(2 * flops_per_direction(N) * ksteps / dtm / 1e9, 1),
                           "tune_variants": tuned_m, "per_pass": per_pass(phm, ksteps)}
        del plan_m

    # N = 1: the kernels one GPU of the 8-GPU runs launches (BASELINE C4 / C5 grids: pencil 2x4; and slab 8), exchange stubbed
    per_gpu_8 = None
    if ngpus == 1 and not args.no_multi_rank_path and N % 8 == 0:
        fill(d_in)
        per_gpu_8 = [per_gpu_kernels(2, 4, 5), per_gpu_kernels(8, 1, 5)]
        # the same plans with the spectrum kept x-contiguous (option spectral_layout = 1: neither x pass touches the point-major layout)
        per_gpu_8_spectral = [per_gpu_kernels(2, 4, 5, {"spectral_layout": 1}), per_gpu_kernels(8, 1, 5, {"spectral_layout": 1})]
        # the reference's own API on the BASELINE grid (execR2C / execC2R, pencil 2 x 4), and what the pipeline depth costs the kernels:
        # every chunk launch of a pass pays ~20 us of launch / drain, which the exchanges hide on real links
        per_gpu_8_more = [per_gpu_kernels(2, 4, 5, None, real=True), per_gpu_kernels(2, 4, 5, {"spectral_layout": 1}, real=True),
                          per_gpu_kernels(2, 4, 5, {"pipeline_chunks": 1}), per_gpu_kernels(2, 4, 5, {"pipeline_chunks": 1}, real=True)]
        fill(d_in)

    # the other decomposition (slab over all ranks next to the BASELINE pencil grid, or the reverse) in the same run
    alt = None
    alt_part = (ngpus, 1) if P2 > 1 else pencil_partition(ngpus)
    if world > 1 and not args.no_alt and alt_part != (P1, P2) and not (args.p1 and args.p2):
        if not aliased:
            del d_back
        plan2, comm2, transport2, work2 = make_plan(*alt_part)
        isz2 = plan2.getInSize()
        n2 = isz2[0] * isz2[1] * isz2[2]
        assert n2 == n_in, "slab and pencil input blocks hold the same number of points"
        a_out = torch.empty(plan2.getDomainSize() // esz, dtype=cdt, device="cuda")
        a_back = d_in if aliased else torch.empty(n2, dtype=cdt, device="cuda")
        if comm2 is not None and transport2 == "torch":
            comm2.register(a_out)
        run_steps(plan2, max(1, min(args.warmup, 3)), a_out, a_back)
        alt_rt = round_trip_error(a_back)
        if aliased:
            fill(d_in)
        plan2.enablePhaseTiming(True)
        dt2, ph2, _ = run_steps(plan2, args.steps, a_out, a_back, collect=True)
        ex2 = sum(ms for name, ms in ph2.items() if "FFT" not in name)
        vol = esz * float(N) ** 3 / ngpus
        a1, a2 = alt_part
        alt = {"decomposition": f"slab P={a1}" if a2 == 1 else f"pencil {a1}x{a2} (BASELINE.json configs)", "transport": transport2,
               "ms_per_step": round(dt2 / args.steps * 1e3, 3), "round_trip_rel_linf": alt_rt,
               "value": round(2 * flops_per_direction(N) * args.steps / dt2 / 1e9, 1),
               "per_pass": per_pass(ph2, args.steps), "exchange_ms_per_step": round(ex2 / args.steps, 3),
               # bytes one GPU sends per step: exchange 1 inside the row group (a2 ranks), exchange 2 inside the column group
               "fft_ms_per_step": round(sum(ms for name, ms in ph2.items() if "FFT" in name) / args.steps, 3),
               "xgmi_bytes_out_per_gpu_per_step": 2.0 * vol * ((a2 - 1) / a2 + (a1 - 1) / a1),
               "xgmi_model_per_transform": xgmi_model(esz, N, ngpus, a1, a2),
               "links_in_use": {"exchange 1": a2 - 1, "exchange 2": a1 - 1}}
        alt["overlap"] = overlap_report(alt["ms_per_step"], alt["fft_ms_per_step"], alt["exchange_ms_per_step"])

    if rank == 0:
        if plain_leg is not None:
            out["config"]["plain_buffers"] = plain_leg
            out["config"]["plain_buffers_ms_per_step"] = plain_leg["ms_per_step"]
            if isinstance(plain_leg.get("unmodified_caller"), dict) and "ms_per_step" in plain_leg["unmodified_caller"]:
                out["config"]["unmodified_caller_ms_per_step"] = plain_leg["unmodified_caller"]["ms_per_step"]
        if relay_leg is not None:
            out["config"]["relay"] = relay_leg
        if multi_rank_path is not None:
            out["config"]["multi_rank_path"] = multi_rank_path
        if per_gpu_8 is not None:
            out["config"]["per_gpu_kernels_8gpu"] = {
                "what": "rank 0's plan of the 8-GPU decompositions run on this GPU with the exchange stubbed out (1/8 of the volume): "
                        "step_ms = forward + inverse as a rank runs them (the chunks of a pass alternate over two compute streams from "
                        "three chunks on), step_ms_one_stream = the same with the chunks serialised on one stream; per_pass / "
                        "kernels_ms_per_step = device spans of the serialised launches (spans of overlapping chunks cannot be summed)",
                "plans": per_gpu_8,
                "spectral_layout_plans": per_gpu_8_spectral,
                "r2c_and_depth_plans": per_gpu_8_more,
                "r2c_and_depth_what": "pencil 2x4: execR2C + execC2R with the reference layout and with spectral_layout = 1, then C2C and R2C with "
                                      "ONE chunk per pass (pipeline_chunks = 1: the kernels without the per-chunk launch cost)",
                "spectral_layout_what": "the same plans with dfft_set_option(plan, 'spectral_layout', 1): the spectrum block is [yo][zs][Nx] "
                                        "(x-contiguous; dfft_get_out_strides), forward x stores and inverse x loads natural lines -- for callers "
                                        "that go forward -> pointwise -> inverse; sizes, starts and exchange tables are the reference's"}
        if alt is not None:
            out["config"]["alt"] = alt
        if ngpus > 1:
            # bytes one GPU puts on xGMI per step (forward + inverse) and the rate the exchange phases
            # reach (device time of the exchange spans, which overlap the kernels when pipelined)
            vol = esz * float(N) ** 3 / ngpus
            out_bytes = 2.0 * vol * ((P1 - 1) / P1 + (P2 - 1) / P2)
            links = max(P1 - 1, 1) if P2 == 1 else None
            xg = {"bytes_out_per_gpu_per_step": out_bytes, "link_peak_GBps": XGMI_LINK_GBS,
                  "achieved_GBps_per_gpu": round(out_bytes / (exch_ms / args.steps * 1e-3) / 1e9, 1) if exch_ms > 0 else None}
            if links:
                xg["links_in_use"] = links
                xg["achieved_GBps_per_link"] = round(xg["achieved_GBps_per_gpu"] / links, 1) if xg["achieved_GBps_per_gpu"] else None
            # the model: per exchange and transform, bytes per link / 153 GB/s, next to the measured span of that exchange
            # (sum over its pipeline chunks, forward and inverse halved)
            model = xgmi_model(esz, N, ngpus, P1, P2)
            for name, m in model.items():
                meas = phases.get(name, 0.0) / args.steps / 2.0
                m["measured_ms"] = round(meas, 3)
                m["measured_GBps_per_link"] = round(m["bytes_per_link"] / (meas * 1e-3) / 1e9, 1) if meas > 0 else None
            xg["per_exchange_per_transform"] = model
            xg["predicted_exchange_ms_per_step"] = round(2.0 * sum(m["predicted_ms"] for m in model.values()), 3)
            out["xgmi"] = xg
            out["overlap"] = overlap_report(ms_per_step, kern_ms / args.steps, exch_ms / args.steps)
        if not args.no_cpu_baseline and ngpus == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
