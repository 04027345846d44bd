"""Worker of tests/test_gpu_multi_device.py: one rank of a one-process-per-GPU run.
usage (under torch.distributed.run): _multi_device_worker.py {slab|pencil|zyx} {rccl|torch}"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import distributedfft_amd as dfft  # noqa: E402
from distributedfft_amd.torch_transport import make_comm  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    kind, transport = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group("nccl")
    P1, P2 = (2, world // 2) if kind == "pencil" else (world, 1)
    comm, name = make_comm(dist, rank, world, P1, P2, transport)
    cls = {"slab": dfft.MPIcuFFT_Slab_Opt1, "pencil": dfft.MPIcuFFT_Pencil_Opt1, "zyx": dfft.MPIcuFFT_Slab_Z_Then_YX}[kind]
    side = torch.cuda.Stream()
    for shape, c2c in (((64, 48, 40), True), ((36, 32, 50), False), ((128, 128, 128), True)):
        plan = cls(dfft.Configurations(), comm, precision="double", rank=rank)
        plan.initFFT(dfft.GlobalSize(*shape), dfft.Partition(P1, P2), allocate=False, c2c=c2c)
        plan.setStream(side.cuda_stream)
        work = torch.empty(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
        plan.setWorkArea(work)
        isz, ist, osz, ost = plan.getInSize(), plan.getInStart(), plan.getOutSize(), plan.getOutStart()
        blk = orc.fill_block(shape, ist, isz, 2 if c2c else 1, seed=5)
        x = torch.from_numpy(np.ascontiguousarray(blk)).cuda()
        out = torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
        back = torch.zeros_like(x)
        if name == "torch":
            comm.register(work)
            comm.register(out)
        torch.cuda.synchronize()
        dist.barrier()
        with torch.cuda.stream(side):
            if c2c:
                plan.execC2C(out, x, dfft.FORWARD)
            else:
                plan.execR2C(out, x)
        torch.cuda.synchronize()
        g = orc.fill_block(shape, (0, 0, 0), shape, 2 if c2c else 1, seed=5)
        want = orc.fft3d_c2c(g, -1) if c2c else orc.fft3d_r2c(g)
        got = out[:osz[0] * osz[1] * osz[2]].cpu().numpy().reshape(osz)
        ref = want[ost[0]:ost[0] + osz[0], ost[1]:ost[1] + osz[1], ost[2]:ost[2] + osz[2]]
        err = np.max(np.abs(got - ref)) / np.max(np.abs(want))
        with torch.cuda.stream(side):
            if c2c:
                plan.execC2C(back, out, dfft.INVERSE)
            else:
                plan.execC2R(back, out)
        torch.cuda.synchronize()
        rt = float((back / float(np.prod(shape)) - x).abs().max()) / 255.0
        t = torch.tensor([err, rt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t[0].item() < 1e-11 and t[1].item() < 1e-10, (kind, shape, c2c, t.tolist())
        del plan
    n = comm.info()[1] if hasattr(comm, "info") else 0
    dist.barrier()
    if rank == 0:
        print(f"MULTI_DEVICE_OK kind={kind} transport={name} rccl_nranks={n}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
