#!/usr/bin/env python3
"""All-to-all bandwidth of the library's exchange step in isolation (the reference's `reference`
executable testcases 1-3 measured the same thing for MPI, tests/src/reference/reference.cu:319-1113).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 tools/exchange_bench.py --size 1024 --transport rccl

Reports, per exchange of the chosen decomposition, bytes sent per GPU, time and GB/s per GPU /
per link.  --backend gloo lets the ranks share one GPU (functional check only)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import distributedfft_amd as dfft  # noqa: E402
from distributedfft_amd.torch_transport import make_comm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--p1", type=int, default=0)
ap.add_argument("--p2", type=int, default=1)
ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "torch"])
ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
torch.cuda.set_device(dev)
if args.backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo")
P1 = args.p1 or world // args.p2
P2 = args.p2
assert P1 * P2 == world
comm, name = make_comm(dist, rank, world, P1, P2, "torch" if args.backend == "gloo" else args.transport)
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), comm, precision="double", rank=rank)
plan.setPipelineChunks(1)
N = args.size
plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(P1, P2), allocate=False, c2c=True)
side = torch.cuda.Stream()
plan.setStream(side.cuda_stream)
send = torch.zeros(plan.getDomainSize(), dtype=torch.uint8, device="cuda")
recv = torch.zeros(plan.getDomainSize(), dtype=torch.uint8, device="cuda")
if name == "torch":
    comm.register(send)
    comm.register(recv)
torch.cuda.synchronize()
for which, n in ((1, P2), (2, P1)):
    if n == 1:
        continue
    sc, _, _, _ = plan.getExchangeTables(which)
    me = rank % P2 if which == 1 else rank // P2
    out_bytes = sum(c for q, c in enumerate(sc) if q != me)
    with torch.cuda.stream(side):
        plan.exchange(which, dfft.FORWARD, send, recv)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            plan.exchange(which, dfft.FORWARD, send, recv)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    if rank == 0:
        print(f"exchange {which} ({n} ranks per group, transport {name}): {out_bytes / 2**20:.1f} MiB out per GPU, "
              f"{dt * 1e3:.3f} ms, {out_bytes / dt / 1e9:.1f} GB/s per GPU, {out_bytes / dt / 1e9 / (n - 1):.1f} GB/s per link")
dist.barrier()
dist.destroy_process_group()
