#!/usr/bin/env python3
"""RCCL / xGMI smoke test of the native transport (distributedfft_amd/csrc/comm.hip), to be run FIRST on a multi-GPU node: it prints
link numbers even if a later FFT leg fails.

    python tools/rccl_smoke.py                 # one GPU: a world of one, the own block through ncclSend / ncclRecv to itself
    python tools/rccl_smoke.py --gpus 8        # starts its own 8 ranks (torch.distributed.run on 127.0.0.1), one per GPU

Prints ONE JSON line on rank 0: RCCL version, ncclCommCount, duplicated communicators, and the measured rate of
  * `pair_shift_GBps[d]`   every rank sends --mib MiB to rank + d and receives from rank - d in one grouped operation (one xGMI link per
                           direction and GPU: the guide's 153 GB/s per link is the yardstick)
  * `alltoall_GBps`        the all-to-all-v over the whole world (what a slab exchange is): bytes out per GPU / time
  * `list_GBps`            the same bytes as a point-to-point schedule with 3 pieces per peer (what a hop of the relay is)
Everything goes through the library's C ABI (dfft_comm_alltoallv / dfft_comm_sendrecv_list); torch.distributed only carries the
ncclUniqueId and the barriers."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mib", type=int, default=256, help="message size per peer")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
                                  "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    import torch

    import distributedfft_amd as dfft
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > ndev:
        sys.exit(f"rccl_smoke: {world} ranks but {ndev} device(s): RCCL needs one GPU per rank")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")      # id broadcast and host barriers only: nothing of RCCL's own is used before the library's transport
        ids = [dfft.Comm.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        uid = ids[0]
    else:
        uid = dfft.Comm.rccl_unique_id()
    t0 = time.perf_counter()
    comm = dfft.Comm.rccl(uid, world, rank)
    t_init = time.perf_counter() - t0
    if world == 1:
        comm.setOption("self_send", 1)
    dup = 0
    try:
        comm.setOption("dup_channel", 3)
        dup = 3
    except Exception as e:   # noqa: BLE001
        print(f"[rank {rank}] dup_channel: {e}", file=sys.stderr, flush=True)
    nb = args.mib << 20
    peers = max(world - 1, 1)
    send = dfft.DeviceBuffer.alloc(nb * world)
    recv = dfft.DeviceBuffer.alloc(nb * world)
    ts, tr = send.tensor(torch.int64), recv.tensor(torch.int64)
    ts.copy_(torch.arange(ts.numel(), device="cuda") + rank * 1000003)
    stream = torch.cuda.Stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(fn):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    out = {"rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "world_size": world, "devices_visible": ndev,
           "ncclCommCount": comm.info()[1], "comm_init_s": round(t_init, 2), "duplicated_communicators": dup, "message_MiB": args.mib,
           "link_peak_GBps": 153.0}
    group = list(range(world))
    # shifts: one partner per direction
    shifts = {}
    for d in (range(1, world) if world > 1 else [0]):
        to, frm = (rank + d) % world, (rank - d) % world
        sc = [nb if q == to else 0 for q in group]
        rc = [nb if q == frm else 0 for q in group]
        sd = [q * nb for q in group]
        dt = timed(lambda: comm.alltoallv(rank, send, sc, sd, recv, rc, sd, group, rank, stream.cuda_stream))
        shifts[str(d)] = round(nb / dt / 1e9, 1)
        if not torch.equal(tr[frm * nb // 8:(frm + 1) * nb // 8][:1024].cpu(), (torch.arange(rank * nb // 8, rank * nb // 8 + 1024) + frm * 1000003)):
            out.setdefault("errors", []).append(f"rank {rank}: shift {d} delivered wrong bytes")
    out["pair_shift_GBps"] = shifts
    sc = [nb] * world
    sd = [q * nb for q in group]
    dt = timed(lambda: comm.alltoallv(rank, send, sc, sd, recv, sc, sd, group, rank, stream.cuda_stream))
    out["alltoall_GBps_out_per_gpu"] = round(nb * peers / dt / 1e9, 1)
    # the same bytes as a schedule: 3 pieces per peer (layers), channel 2 (a duplicated communicator where there is one)
    third = (nb // 3) & ~255
    cuts = [(0, third), (third, third), (2 * third, nb - 2 * third)]
    plist = [q for q in group if q != rank] or [rank]
    sends = [(q, layer, send.address + q * nb + off, ln) for q in plist for layer, (off, ln) in enumerate(cuts)]
    recvs = [(q, layer, recv.address + q * nb + off, ln) for q in plist for layer, (off, ln) in enumerate(cuts)]
    comm.setOption("test_channel", 2)
    dt = timed(lambda: comm.sendrecvList(rank, sends, recvs, 3, stream.cuda_stream))
    comm.setOption("test_channel", 0)
    out["list_GBps_out_per_gpu"] = round(nb * len(plist) / dt / 1e9, 1)
    out["transport_counters"] = comm.counters()
    if dist is not None:
        errs = [None] * world
        dist.all_gather_object(errs, out.get("errors", []))
        out["errors"] = [e for lst in errs for e in lst]
    if rank == 0:
        print(json.dumps(out), flush=True)
    barrier()
    del ts, tr
    send.free(); recv.free()
    comm.setOption("dup_channel", 0)
    comm.destroy()
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(1 if out.get("errors") else 0)


if __name__ == "__main__":
    main()
