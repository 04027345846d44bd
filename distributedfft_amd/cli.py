"""Command-line test driver with the reference executables' flags and outputs.

    python -m distributedfft_amd.cli pencil -nx 256 -ny 256 -nz 256 -p1 2 -p2 4 -o 1 -t 3 -i 5 -d -c
    python -m distributedfft_amd.cli slab   -nx 256 -ny 256 -nz 256 -t 4 -d -p 4

mirrors `tests/src/pencil/main.cpp:26-236` / `tests/src/slab/main.cpp` (flag names, testcases 0-4,
`--fft-dim`), prints `Result (avg)` / `Result (max)` like `random_dist_3D.cu:663-666`, and appends
a timer CSV in the reference's format (`src/timer.cpp:58-101`: header row of rank ids, then one
`section,t_rank0,t_rank1,...` block per timed iteration, cumulative milliseconds) under
`<benchmark_dir>/{pencil,slab_default}/` with the reference's file-name scheme
(`src/pencil/mpicufft_pencil_opt1.cpp:50-55`).

Ranks: by default the P1*P2 ranks are virtual ranks sharing GPU 0 (one host thread each), the
analogue of `mpiexec -n P` with `cudaSetDevice(rank % dev_count)` on a 1-GPU box.  Under
`torch.distributed.run` (WORLD_SIZE > 1) every process is ONE rank on GPU `LOCAL_RANK % device_count`
-- the reference's `mpiexec -n P ./pencil ...` (tests/src/pencil/main.cpp:194-229):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m distributedfft_amd.cli pencil -nx 1024 -ny 1024 -nz 1024 -p1 2 -p2 4 -o 1 -t 3 -i 20 -w 10 -d -c

The exchange then runs over RCCL (`--backend nccl`, the library's own communicator or torch's, chosen like in
bench.py) or, with `--backend gloo`, through host memory so that several ranks may share one GPU.  Error norms are
reduced over the ranks, rank 0 prints them and writes the timer CSV with one column per rank.

Test utilities only (random fill, the Laplacian multiplier, error norms use torch ops); the
transforms go through the C ABI.
"""
import argparse
import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

COMM = {"Peer2Peer": 0, "All2All": 1}
SEND = {"Sync": 0, "Streams": 1, "MPI_Type": 2}
SECTIONS_F = ["1D FFT Z-Direction", "First Transpose (Finished All2All)", "1D FFT Y-Direction",
              "Second Transpose (Finished All2All)", "1D FFT X-Direction"]
SECTIONS_B = ["1D FFT X-Direction", "First Transpose (Finished All2All)", "1D FFT Y-Direction",
              "Second Transpose (Finished All2All)", "1D FFT Z-Direction"]


def parse(argv):
    ap = argparse.ArgumentParser(prog="distributedfft_amd.cli", allow_abbrev=False)
    ap.add_argument("mode", choices=["pencil", "slab"])
    ap.add_argument("--input-dim-x", "-nx", type=int, required=True)
    ap.add_argument("--input-dim-y", "-ny", type=int, required=True)
    ap.add_argument("--input-dim-z", "-nz", type=int, required=True)
    ap.add_argument("--partition1", "-p1", type=int, default=0)
    ap.add_argument("--partition2", "-p2", type=int, default=1)
    ap.add_argument("--partition", "-p", type=int, default=0, help="slab: number of ranks")
    ap.add_argument("--comm-method1", "-comm1", "--comm-method", "-comm", default="All2All", choices=list(COMM))
    ap.add_argument("--send-method1", "-snd1", "--send-method", "-snd", default="Sync", choices=list(SEND))
    ap.add_argument("--comm-method2", "-comm2", default="All2All", choices=list(COMM))
    ap.add_argument("--send-method2", "-snd2", default="Sync", choices=list(SEND))
    ap.add_argument("--testcase", "-t", type=int, default=0, choices=[0, 1, 2, 3, 4])
    ap.add_argument("--opt", "-o", type=int, default=0, choices=[0, 1])
    ap.add_argument("--fft-dim", "-f", type=int, default=3, choices=[1, 2, 3])
    ap.add_argument("--iterations", "-i", type=int, default=1)
    ap.add_argument("--warmup-rounds", "-w", type=int, default=0)
    ap.add_argument("--cuda_aware", "-c", action="store_true")
    ap.add_argument("--double_prec", "-d", action="store_true")
    ap.add_argument("--benchmark_dir", "-b", default="../benchmarks")
    ap.add_argument("--sequence", "-s", default="ZY_Then_X", choices=["ZY_Then_X", "Z_Then_YX", "Y_Then_ZX"],
                    help="slab only (tests/src/slab/main.cpp:138-140)")
    ap.add_argument("--complex", action="store_true", help="extension: complex-to-complex instead of R2C/C2R")
    ap.add_argument("--spectral-layout", type=int, default=0, choices=[0, 1],
                    help="extension (pencil and default slab classes): 1 = the spectrum block stays x-contiguous, [yo][zs][Nx] "
                         "(dfft_set_option 'spectral_layout'); the testcases index it through the plan's strides")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="one-process-per-rank runs: torch.distributed backend (gloo: ranks may share a GPU)")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "torch"], help="one-process-per-rank runs, backend nccl")
    return ap.parse_args(argv)


class Rank:
    def __init__(self, args, comm, rank, P1, P2, register=None, own_stream=False):
        import torch

        import distributedfft_amd as dfft
        self.torch, self.dfft, self.args, self.rank = torch, dfft, args, rank
        prec = "double" if args.double_prec else "float"
        self.rdt = torch.float64 if args.double_prec else torch.float32
        self.cdt = torch.complex128 if args.double_prec else torch.complex64
        self.esz = 16 if args.double_prec else 8
        cfg = dfft.Configurations(args.cuda_aware, args.warmup_rounds, COMM[args.comm_method1], SEND[args.send_method1],
                                  args.benchmark_dir, COMM[args.comm_method2], SEND[args.send_method2])
        kind = {("pencil", 0): dfft.MPIcuFFT_Pencil, ("pencil", 1): dfft.MPIcuFFT_Pencil_Opt1,
                ("slab", 0): dfft.MPIcuFFT_Slab, ("slab", 1): dfft.MPIcuFFT_Slab_Opt1}[(args.mode, args.opt)]
        if args.mode == "slab" and args.sequence == "Z_Then_YX":
            kind = dfft.MPIcuFFT_Slab_Z_Then_YX if args.opt == 0 else dfft.MPIcuFFT_Slab_Z_Then_YX_Opt1
        if args.mode == "slab" and args.sequence == "Y_Then_ZX":
            kind = dfft.MPIcuFFT_Slab_Y_Then_ZX
        self.plan = kind(cfg, comm, precision=prec, rank=rank)
        if getattr(args, "spectral_layout", 0):
            self.plan.setOption("spectral_layout", args.spectral_layout)
        t0 = time.perf_counter()
        self.N = (args.input_dim_x, args.input_dim_y, args.input_dim_z)
        # one process per rank: the plan runs on a dedicated torch stream (the torch transport's collectives order
        # themselves against torch's current stream) and exchanges only registered tensors
        self.side = torch.cuda.Stream() if register is not None or own_stream else None
        self.plan.initFFT(dfft.GlobalSize(*self.N), dfft.Partition(P1, P2), register is None, c2c=args.complex)
        if self.side is not None:
            self.plan.setStream(self.side.cuda_stream)
        if register is not None:
            self.work = dfft.DeviceBuffer.alloc(self.plan.getWorkSizeDevice()).tensor(torch.uint8)      # the library's default backing
            self.plan.setWorkArea(self.work)
            register(self.work)
        self.plan.enablePhaseTiming(True)
        self.init_ms = (time.perf_counter() - t0) * 1e3
        self.isz, self.ist = self.plan.getInSize(), self.plan.getInStart()
        self.osz, self.ost = self.plan.getOutSize(), self.plan.getOutStart()
        # `out` from the library's allocator (dfft_malloc(DFFT_CHUNK_DEFAULT)): the caller owns it as in the reference, on the
        # backing the scatter passes run faster on
        self.out = dfft.DeviceBuffer.alloc(self.plan.getDomainSize()).tensor(self.cdt)
        self.out.zero_()
        if register is not None:
            register(self.out)
        self.timings = []

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def rand_input(self):
        """uniform (0,1] * 255 like initializeRandArray (tests/src/pencil/base.cu:39-58), but seeded"""
        t = self.torch
        g = t.Generator(device="cuda")
        g.manual_seed(1000 + self.rank)
        n = self.isz[0] * self.isz[1] * self.isz[2]
        if self.args.complex:
            return t.view_as_complex(t.rand((n, 2), dtype=self.rdt, device="cuda", generator=g) * 255).reshape(self.isz)
        return (t.rand(n, dtype=self.rdt, device="cuda", generator=g) * 255).reshape(self.isz)

    def forward(self, x, d=3):
        with self._on_stream():
            if self.args.complex:
                self.plan.execC2C(self.out, x, self.dfft.FORWARD, d=d)
            else:
                self.plan.execR2C(self.out, x, d)
        self.record(self.dfft.FORWARD)

    def inverse(self, y, d=3):
        with self._on_stream():
            if self.args.complex:
                self.plan.execC2C(y, self.out, self.dfft.INVERSE, d=d)
            else:
                self.plan.execC2R(y, self.out, d)
        self.record(self.dfft.INVERSE)

    def record(self, direction):
        ph = self.plan.getPhaseTimes(direction)
        names = SECTIONS_F if direction == self.dfft.FORWARD else SECTIONS_B
        if self.args.sequence == "Y_Then_ZX":      # phases: y pass, exchange, x pass, -, z pass
            names = ["1D FFT Y-Direction", "Transpose (Finished Receive)", "1D FFT X-Direction", "-", "1D FFT Z-Direction"]
        elif self.args.sequence == "Z_Then_YX":    # one exchange only (phase 1 forward, phase 3 inverse)
            names = [n.replace("First Transpose (Finished All2All)", "Transpose (Finished Receive)")
                      .replace("Second Transpose (Finished All2All)", "Transpose (Finished Receive)") for n in names]
        cum, rows = 0.0, []
        for (_, ms), name in zip(ph, names):
            cum += ms
            rows.append((name, cum))
        rows.append(("Run complete", cum))
        self.timings.append(rows)

    def spectrum(self):
        """my spectrum block as an (Nx, yo, zs) view of `out`, whatever the plan's spectral layout (dfft_get_out_strides)"""
        return self.plan.spectrumView(self.out)

    def laplacian_multiplier(self):
        """derivativeCoefficients (tests/src/pencil/random_dist_3D.cu:98-121) on my output block"""
        t = self.torch
        Nx, Ny, Nz = self.N
        blk = self.spectrum()

        def wrapped(idx, N, half=False):
            k = t.where(idx < N // 2, idx, t.where(idx > N // 2, N - idx, t.zeros_like(idx)))
            if half:
                k = t.where(idx < N // 2, idx, t.zeros_like(idx))
            return k.to(t.float64)

        k1 = wrapped(t.arange(Nx, device="cuda"), Nx).reshape(-1, 1, 1)
        k2 = wrapped(t.arange(self.ost[1], self.ost[1] + self.osz[1], device="cuda"), Ny).reshape(1, -1, 1)
        k3 = wrapped(t.arange(self.ost[2], self.ost[2] + self.osz[2], device="cuda"), Nz, half=not self.args.complex).reshape(1, 1, -1)
        # the reference's arithmetic: x * scale / sqrtf(Nx*Ny*Nz) -- a SINGLE-precision root of the int product (:117-118); it is why
        # its own runs print 1.91723e-05 at 128^3 (tests/golden/ref_testcase4_results.json)
        root = float(t.sqrt(t.tensor(float(Nx * Ny * Nz), dtype=t.float32)))
        scale = -(k1 ** 2 + k2 ** 2 + k3 ** 2)
        rd = blk.real.dtype
        t.view_as_real(blk).copy_((t.view_as_real(blk).to(t.float64) * scale.unsqueeze(-1) / root).to(rd))


class _Timings:      # what write_csv needs of a rank (the rank objects themselves, or what other processes sent to rank 0)
    def __init__(self, init_ms, timings):
        self.init_ms, self.timings = init_ms, timings


def write_csv(args, ranks, P1, P2):
    sub = "pencil" if args.mode == "pencil" else ({"Z_Then_YX": "slab_z_then_yx", "Y_Then_ZX": "slab_y_then_zx"}.get(args.sequence, "slab_default"))
    d = os.path.join(args.benchmark_dir, sub)
    os.makedirs(d, exist_ok=True)
    if args.mode == "pencil":
        name = (f"test_{args.opt}_{COMM[args.comm_method1]}_{SEND[args.send_method1]}_{COMM[args.comm_method2]}_"
                f"{SEND[args.send_method2]}_{args.input_dim_x}_{args.input_dim_y}_{args.input_dim_z}_"
                f"{int(args.cuda_aware)}_{P1}_{P2}.csv")
    else:
        name = (f"test_{args.opt}_{COMM[args.comm_method1]}_{SEND[args.send_method1]}_{args.input_dim_x}_"
                f"{args.input_dim_y}_{args.input_dim_z}_{int(args.cuda_aware)}_{P1}.csv")
    path = os.path.join(d, name)
    new = not os.path.exists(path)
    with open(path, "a") as f:
        if new:
            f.write("," + "".join(f"{r}," for r in range(len(ranks))))
        skip = args.warmup_rounds * (2 if args.testcase in (3, 4) else 1)   # forward + inverse blocks per iteration
        per_iter = len(ranks[0].timings)
        for it in range(per_iter):
            if it < skip:
                continue
            f.write("\n")
            if it == skip:
                f.write("init," + "".join(f"{rk.init_ms}," for rk in ranks) + "\n")
            for s, (nm, _) in enumerate(ranks[0].timings[it]):
                f.write(nm + "," + "".join(f"{rk.timings[it][s][1]}," for rk in ranks) + "\n")
    return path


def run(argv=None):
    args = parse(argv if argv is not None else sys.argv[1:])
    if args.sequence != "ZY_Then_X" and args.mode != "slab":
        raise SystemExit("--sequence applies to the slab decomposition")
    if args.sequence == "Y_Then_ZX" and args.testcase not in (0, 1):
        raise SystemExit("sequence Y_Then_ZX is forward only (testcases 0 and 1), as in the reference")
    if args.sequence != "ZY_Then_X" and args.fft_dim != 3:
        raise SystemExit("--fft-dim 1|2 is defined for the pencil classes only")
    import torch

    import distributedfft_amd as dfft
    if args.mode == "slab":
        P1, P2 = (args.partition or args.partition1 or 1), 1
    else:
        P1, P2 = (args.partition1 or 1), args.partition2
    P = P1 * P2
    nproc = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if nproc > 1:
        # one process per rank (the reference's mpiexec -n P): this process is rank RANK of P
        import torch.distributed as dist

        from distributedfft_amd.torch_transport import make_comm
        if nproc != P:
            raise SystemExit(f"{nproc} processes but a {P1}x{P2} partition")
        me = int(os.environ["RANK"])
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(args.backend)
        comm, transport = make_comm(dist, me, nproc, P1, P2, "torch" if args.backend == "gloo" else args.transport)
        register = comm.register if transport == "torch" else None
        ranks = [Rank(args, comm, me, P1, P2, register=register, own_stream=True)]
        pool = None
    else:
        me = 0
        world = dfft.Comm.local(P) if P > 1 else None
        ranks = [Rank(args, world, r, P1, P2) for r in range(P)]
        pool = ThreadPoolExecutor(P)
    root = me == 0

    def each(fn):
        if pool is None:
            fn(ranks[0])
            torch.cuda.synchronize()
            dist.barrier()
        else:
            list(pool.map(fn, ranks))
            torch.cuda.synchronize()

    def gsum(v):      # sum / max over all ranks (values of this process's ranks are already combined)
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def gmax(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    Nx, Ny, Nz = ranks[0].N
    n3 = float(Nx) * Ny * Nz
    d = args.fft_dim
    iters = args.iterations + args.warmup_rounds
    result = {}
    if args.testcase in (0, 2):
        xs = {rk.rank: rk.rand_input() for rk in ranks}
        ys = {rk.rank: torch.zeros_like(xs[rk.rank]) for rk in ranks}
        torch.cuda.synchronize()
        if args.testcase == 2:
            each(lambda rk: rk.forward(xs[rk.rank], d))
            for rk in ranks:
                rk.timings.clear()
        for _ in range(max(iters, 1)):
            if args.testcase == 0:
                each(lambda rk: rk.forward(xs[rk.rank], d))
            else:
                each(lambda rk: rk.inverse(ys[rk.rank], d))
    elif args.testcase == 1:
        # distributed == single device (tests/src/pencil/random_dist_3D.cu:229-504); the coordinator's
        # one-GPU transform is a single-rank plan of this library on the same device
        xs = {rk.rank: rk.rand_input() for rk in ranks}
        full = torch.zeros(ranks[0].N, dtype=xs[ranks[0].rank].dtype, device="cuda")
        for rk in ranks:
            full[rk.ist[0]:rk.ist[0] + rk.isz[0], rk.ist[1]:rk.ist[1] + rk.isz[1], :] = xs[rk.rank]
        if dist is not None:      # assemble the global input on every process (testcase 1 runs on small grids)
            buf = full.cpu() if args.backend == "gloo" else full
            # gloo (and some RCCL builds) have no complex all_reduce: reduce the interleaved real view
            dist.all_reduce(torch.view_as_real(buf) if buf.is_complex() else buf, op=dist.ReduceOp.SUM)
            full = buf.cuda()
        single = Rank(args, None, 0, 1, 1)
        torch.cuda.synchronize()
        single.forward(full.contiguous(), 3)
        each(lambda rk: rk.forward(xs[rk.rank], 3))
        ref = single.spectrum()
        tot = 0.0
        for rk in ranks:
            blk = rk.spectrum()
            tot += float((blk - ref[:, rk.ost[1]:rk.ost[1] + rk.osz[1], rk.ost[2]:rk.ost[2] + rk.osz[2]]).abs().sum())
        tot = gsum(tot)
        result = {"sum": tot}
        if root:
            print(f"Result {tot}")
    elif args.testcase == 3:
        xs = {rk.rank: rk.rand_input() for rk in ranks}
        ys = {rk.rank: torch.zeros_like(xs[rk.rank]) for rk in ranks}
        torch.cuda.synchronize()
        for _ in range(max(iters, 1)):
            each(lambda rk: rk.forward(xs[rk.rank], 3))
            each(lambda rk: rk.inverse(ys[rk.rank], 3))
            diffs = [(ys[rk.rank] - n3 * xs[rk.rank]).abs() for rk in ranks]   # differenceInv(inv, in, n, N^3), :650
            s, m = gsum(sum(float(v.sum()) for v in diffs)), gmax(max(float(v.max()) for v in diffs))
            result = {"avg": s / n3, "max": m}
            if root:
                print(f"Result (avg): {s / n3}")
                print(f"Result (max): {m}")
    elif args.testcase == 4:
        us, ders = {}, {}
        for rk in ranks:
            ax = [torch.arange(rk.ist[a], rk.ist[a] + rk.isz[a], device="cuda", dtype=torch.float64) for a in range(3)]
            u = (torch.sin(2 * math.pi * ax[0] / Nx).reshape(-1, 1, 1) * torch.sin(2 * math.pi * ax[1] / Ny).reshape(1, -1, 1)
                 * torch.sin(2 * math.pi * ax[2] / Nz).reshape(1, 1, -1)).to(rk.rdt)
            us[rk.rank] = u.to(rk.cdt) if args.complex else u
            ders[rk.rank] = -3.0 * math.sqrt(n3) * us[rk.rank]
        ys = {rk.rank: torch.zeros_like(us[rk.rank]) for rk in ranks}
        torch.cuda.synchronize()
        for _ in range(max(iters, 1)):
            each(lambda rk: rk.forward(us[rk.rank], 3))
            for rk in ranks:
                rk.laplacian_multiplier()
            torch.cuda.synchronize()
            each(lambda rk: rk.inverse(ys[rk.rank], 3))
            diffs = [(ys[rk.rank] - ders[rk.rank]).abs() for rk in ranks]
            s, m = gsum(sum(float(v.sum()) for v in diffs)), gmax(max(float(v.max()) for v in diffs))
            result = {"avg": s / n3, "max": m}
            if root:
                print(f"Result (avg): {s / n3}")
                print(f"Result (max): {m}")
    if dist is None:
        result["csv"] = write_csv(args, ranks, P1, P2)
        pool.shutdown()
    else:
        # the reference gathers the timings on rank 0 (MPI_Gatherv in src/timer.cpp:58-101) and writes one column per rank
        gathered = [None] * nproc if root else None
        dist.gather_object((ranks[0].init_ms, ranks[0].timings), gathered, dst=0)
        if root:
            result["csv"] = write_csv(args, [_Timings(i, t) for i, t in gathered], P1, P2)
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    r = run()
    if int(os.environ.get("RANK", "0")) == 0:
        print("cli result:", {k: v for k, v in r.items()})
