#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

Metric (BASELINE.json): 3-D FFT GFLOP/s with the 5*N^3*log2(N^3) convention, forward + inverse,
1024^3 fp64 complex, at 1/2/4/8 GPUs.  A "step" is one forward plus one inverse transform of the
resident grid (the reference's testcase 0 + testcase 2 back to back,
tests/src/pencil/random_dist_3D.cu:154-227, :506-579), inputs already in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 2

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- achieved HBM GB/s of the dominant kernel (the axis-pass kernel), from
                  algorithmic bytes per launch / HIP-event duration on the launch stream
  cpu_baseline -- the CPU oracle (a port: the reference has no CPU path) timed on this host
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024, help="cube edge (BASELINE: 1024)")
    ap.add_argument("--precision", default="double", choices=["double", "float"])
    ap.add_argument("--p1", type=int, default=0)
    ap.add_argument("--p2", type=int, default=0)
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "torch"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo lets several ranks share one GPU (functional test)")
    ap.add_argument("--decomp", default="auto", choices=["auto", "slab", "pencil"],
                    help="auto = slab on one xGMI node (every GPU pair has its own link), pencil = BASELINE 2x4 / 2x2")
    ap.add_argument("--no-alt", action="store_true", help="skip the alternative-decomposition measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=512, help="cube edge of the CPU-baseline sample")
    return ap.parse_args()


def pencil_partition(n):
    # BASELINE.json configs: 1 GPU local passes; 2 GPUs slab; 8 GPUs 2x4 pencil; 4 GPUs 2x2 pencil
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}.get(n, (n, 1))


def choose_partition(n, decomp):
    """On one xGMI node every GPU pair has a private link, so a P-way all-to-all drives P-1
    links at once: slab (one exchange over all P ranks) moves (P-1)/P of the volume once over
    P-1 links, pencil 2x4 moves 3/4 over 3 links and then 1/2 over a single link.
    auto therefore picks slab; the BASELINE-named pencil grid is measured next to it."""
    if decomp == "pencil":
        return pencil_partition(n)
    return (n, 1)


def cpu_baseline(n):
    """Times the CPU oracle (oracle/dfft_oracle.c, OpenMP over lines) on an n^3 fp64 complex
    forward+inverse.  kind = 'port': the reference has no CPU implementation to build."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as orc
    L = orc.lib()
    g = orc.fill_block((n, n, n), (0, 0, 0), (n, n, n), 2, seed=20260921)
    ptr = g.ctypes.data_as(C.c_void_p)
    L.orc_fft3d_c2c(ptr, n, n, n, -1)      # warm-up (thread pool, page faults)
    L.orc_fft3d_c2c(ptr, n, n, n, +1)
    g /= float(n) ** 3
    iters, t0 = 0, time.perf_counter()
    while True:
        L.orc_fft3d_c2c(ptr, n, n, n, -1)
        L.orc_fft3d_c2c(ptr, n, n, n, +1)
        g /= float(n) ** 3
        iters += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or iters >= 20:
            break
    flops = 2 * 5.0 * n ** 3 * math.log2(float(n) ** 3)
    return {"value": round(flops * iters / dt / 1e9, 3), "unit": "GFLOP/s", "cores": orc.num_threads(),
            "kind": "port",
            "sample": f"{n}^3 fp64 complex forward+inverse x{iters} ({dt:.1f} s), oracle/dfft_oracle.c, "
                      f"host has {os.cpu_count()} cores"}


def main():
    args = parse()
    import torch

    import distributedfft_amd as dfft

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ngpus = args.gpus
    if world != ngpus:
        if world == 1 and ngpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        ngpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
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

    def make_plan(P1, P2):
        comm, transport = None, "none"
        if world > 1:
            from distributedfft_amd.torch_transport import make_comm
            comm, transport = make_comm(dist, rank, world, P1, P2, tmode_box[0])
        kind = dfft.MPIcuFFT_Slab_Opt1 if P2 == 1 and ngpus > 1 else dfft.MPIcuFFT_Pencil_Opt1
        plan = kind(dfft.Configurations(), comm, precision=prec, rank=rank)
        plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(P1, P2), allocate=False, c2c=True)
        plan.setStream(stream)
        if comm is not None and transport == "torch":
            # the torch transport maps raw pointers back to tensors, so the work area must be one
            work = torch.empty(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
            plan.setWorkArea(work)
            comm.register(work)
        else:
            work = None
            plan.setWorkArea(None)       # library-owned (hipMalloc), like the reference's allocate = true
        return plan, comm, transport, work

    plan, comm, transport, work = make_plan(P1, P2)
    domain = plan.getDomainSize()

    # synthetic input: this rank's block of a uniform[0,255) complex grid (the reference scales
    # cuRAND uniforms by 255, tests/src/pencil/base.cu:45-53), generated on the device
    isz = plan.getInSize()
    n_in = isz[0] * isz[1] * isz[2]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(20260921 + rank)
    d_in = torch.empty(n_in, dtype=cdt, device="cuda")
    v = torch.view_as_real(d_in)
    chunk = 1 << 26
    for o in range(0, n_in, chunk):
        e = min(n_in, o + chunk)
        v[o:e] = torch.rand((e - o, 2), dtype=rdt, device="cuda", generator=gen) * 255.0
    d_out = torch.empty(domain // esz, dtype=cdt, device="cuda")
    d_back = torch.empty(n_in, dtype=cdt, device="cuda")
    if comm is not None and transport == "torch":
        comm.register(d_out)
    ref_sample = d_in[:4096].clone()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        with torch.cuda.stream(side):
            plan.execC2C(d_out, d_in, dfft.FORWARD)       # blocking, like the reference's exec
            plan.execC2C(d_back, d_out, dfft.INVERSE)
        torch.cuda.synchronize()

    def warm_and_check():
        for _ in range(args.warmup):
            step()
        # round-trip check on the warm-up result (reference testcase 3), worst rank
        if args.warmup == 0:
            return None
        diff = (d_back[:4096] / float(N) ** 3 - ref_sample).abs().max() / ref_sample.abs().max()
        full = (d_back / float(N) ** 3 - d_in).abs().max() / d_in.abs().max()
        e = torch.maximum(diff, full).reshape(1).to(torch.float64)
        if dist is not None:
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
        return float(e)

    rt_err = warm_and_check()
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
        rt_err = warm_and_check()

    plan.enablePhaseTiming(True)
    kern_ms, kern_launches = 0.0, 0
    exch_ms = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        with torch.cuda.stream(side):
            plan.execC2C(d_out, d_in, dfft.FORWARD)
            ph_f = plan.getPhaseTimes(dfft.FORWARD)
            plan.execC2C(d_back, d_out, dfft.INVERSE)
            ph_b = plan.getPhaseTimes(dfft.INVERSE)
        for name, ms in ph_f + ph_b:
            if "FFT" in name:
                kern_ms += ms
                kern_launches += 1
            else:
                exch_ms += ms
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # max over ranks of the per-phase device times as well
        t = torch.tensor([kern_ms, exch_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kern_ms, exch_ms = float(t[0]), float(t[1])

    # the BASELINE-named pencil grid (2x2 / 2x4) measured next to the chosen decomposition
    alt = None
    alt_part = pencil_partition(ngpus)
    if world > 1 and not args.no_alt and alt_part != (P1, P2) and not (args.p1 and args.p2):
        del d_back
        plan2, comm2, transport2, work2 = make_plan(*alt_part)
        isz2 = plan2.getInSize()
        n2 = isz2[0] * isz2[1] * isz2[2]
        a_in = d_in[:n2] if n2 <= d_in.numel() else torch.zeros(n2, dtype=cdt, device="cuda")
        a_out = torch.empty(plan2.getDomainSize() // esz, dtype=cdt, device="cuda")
        a_back = torch.empty(n2, dtype=cdt, device="cuda")
        if comm2 is not None and transport2 == "torch":
            comm2.register(a_out)
        with torch.cuda.stream(side):
            for _ in range(max(1, args.warmup)):
                plan2.execC2C(a_out, a_in, dfft.FORWARD)
                plan2.execC2C(a_back, a_out, dfft.INVERSE)
        alt_rt = float((a_back / float(N) ** 3 - a_in).abs().max() / a_in.abs().max())
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            for _ in range(args.steps):
                plan2.execC2C(a_out, a_in, dfft.FORWARD)
                plan2.execC2C(a_back, a_out, dfft.INVERSE)
        barrier()
        dt2 = time.perf_counter() - t0
        t = torch.tensor([dt2], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt2 = float(t.item())
        alt = {"decomposition": f"pencil {alt_part[0]}x{alt_part[1]} (BASELINE.json configs)", "transport": transport2,
               "ms_per_step": round(dt2 / args.steps * 1e3, 3), "round_trip_rel_linf": alt_rt,
               "value": round(2 * 5.0 * float(N) ** 3 * math.log2(float(N) ** 3) * args.steps / dt2 / 1e9, 1)}

    flops_step = 2 * 5.0 * float(N) ** 3 * math.log2(float(N) ** 3)
    ms_per_step = dt / args.steps * 1e3
    value = flops_step * args.steps / dt / 1e9

    # roofline of the dominant kernel: fft_pass_kernel.  One axis pass (one launch on a single
    # GPU; `pipeline_chunks` launches when the pass is pipelined against an exchange) reads the
    # local volume once and writes it once: algorithmic bytes = 2 * esz * N^3 / n_gpus
    # (SURVEY.md 8d).  Duration = HIP events around the launches on the launch stream.
    bytes_launch = 2.0 * esz * float(N) ** 3 / ngpus
    avg_ms = kern_ms / max(kern_launches, 1)
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "dfft::fft_pass_kernel", "avg_launch_ms": round(avg_ms, 4),
                "launches_timed": kern_launches, "alg_bytes_per_launch": bytes_launch,
                "launches_per_pass": plan.getPipelineChunks() if ngpus > 1 else 1}

    # HBM bytes per launch from the committed PMC profile of this kernel on this workload
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read correction applied)
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
        if ngpus == 1 and N == 1024 and prec == "double":
            roofline["traffic"] = pm["hbm_bytes_per_launch"]
            roofline["traffic_source"] = "profiles/r1_pmc_traffic.json"
    except Exception:   # noqa: BLE001
        pass

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
                       "pipeline_chunks": plan.getPipelineChunks()},
            "round_trip_rel_linf": rt_err,
            "roofline": roofline,
        }
        if alt is not None:
            out["config"]["alt"] = alt
        if ngpus > 1:
            # bytes one GPU puts on xGMI per step (forward + inverse) and the rate the exchange phases
            # reach (device time of the exchange spans, which overlap the kernels when pipelined)
            vol = esz * float(N) ** 3 / ngpus
            out_bytes = 2.0 * vol * ((P1 - 1) / P1 + (P2 - 1) / P2)
            links = max(P1 - 1, 1) if P2 == 1 else None
            xg = {"bytes_out_per_gpu_per_step": out_bytes,
                  "achieved_GBps_per_gpu": round(out_bytes / (exch_ms / args.steps * 1e-3) / 1e9, 1) if exch_ms > 0 else None}
            if links:
                xg["links_in_use"] = links
                xg["achieved_GBps_per_link"] = round(xg["achieved_GBps_per_gpu"] / links, 1) if xg["achieved_GBps_per_gpu"] else None
            out["xgmi"] = xg
        if not args.no_cpu_baseline and ngpus == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
