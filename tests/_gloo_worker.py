"""Worker for tests/test_cpu_multiprocess.py: one process per rank over gloo (no GPU).

Runs the REFERENCE's opt1 data flow (oracle 1-D FFTs, reference buffer layouts) but moves every
block between ranks with the product's own exchange: plan tables from libdfft_amd.so
(dfft_get_exchange_tables) + dfft_exchange() + the torch.distributed callback transport.  This
covers the N > 1 host path -- grid coordinates, row/column groups, byte tables, pointer
registration, forward and inverse table swap -- on CPU.  The HIP kernels themselves are covered
by the -m gpu tests."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import distributedfft_amd as dfft  # noqa: E402
from distributedfft_amd.torch_transport import TorchComm  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    P1, P2 = int(sys.argv[1]), int(sys.argv[2])
    shape = tuple(int(v) for v in sys.argv[3].split("x"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tc = TorchComm(dist, rank, world, P1, P2)
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), tc, precision="double", rank=rank)
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(P1, P2), allocate=False, c2c=True)
    assert plan.getRank() == rank and plan.getWorldSize() == world
    Nx, Ny, Nz = shape
    i, j = rank // P2, rank % P2
    opl = orc.PencilPlan(Nx, Ny, Nz, P1, P2, True)
    assert (plan.getInSize(), plan.getInStart()) == opl.in_block(rank)
    assert (plan.getOutSize(), plan.getOutStart()) == opl.out_block(rank)
    nel = plan.getDomainSize() // 16
    A = torch.zeros(nel, dtype=torch.complex128)
    B = torch.zeros(nel, dtype=torch.complex128)
    tc.register(A)
    tc.register(B)
    a, b = A.numpy(), B.numpy()

    isz, ist = plan.getInSize(), plan.getInStart()
    xs, ys = isz[0], isz[1]
    blk = orc.fill_block(shape, ist, isz, 2, seed=21)
    # z-FFT -> [z][x][y]   (mpicufft_pencil_opt1.cpp:165-168)
    z = orc.fft1d(blk.reshape(xs * ys, Nz), -1)
    a[:xs * ys * Nz] = z.T.ravel()
    send_copy = a.copy()
    if P2 > 1:
        plan.exchange(1, dfft.FORWARD, A, B)
    else:
        b[:] = a          # single-member group: the product's next pass reads the buffer in place
    osz, ost = plan.getOutSize(), plan.getOutStart()
    zs, yo = osz[2], osz[1]
    # unpack [zs][xs][ys_p] -> [zs][xs][Ny]   (:788-800)
    _, _, rc, rd = plan.getExchangeTables(1)
    temp = np.zeros((zs, xs, Ny), dtype=np.complex128)
    for p in range(P2):
        ysp, y0 = opl.in_block(i * P2 + p)[0][1], opl.in_block(i * P2 + p)[1][1]
        assert rc[p] == 16 * xs * ysp * zs
        temp[:, :, y0:y0 + ysp] = b[rd[p] // 16: rd[p] // 16 + xs * ysp * zs].reshape(zs, xs, ysp)
    # inverse direction of the same exchange must return every block to its sender
    if P2 > 1:
        A.zero_()
        plan.exchange(1, dfft.INVERSE, B, A)
        assert np.array_equal(a[:xs * ys * Nz], send_copy[:xs * ys * Nz]), "inverse exchange 1 is not the mirror"
    # y-FFT -> [y][zs][xs]   (:177-180)
    y = orc.fft1d(temp.reshape(zs * xs, Ny), -1)
    a[:zs * xs * Ny] = y.T.ravel()
    if P1 > 1:
        plan.exchange(2, dfft.FORWARD, A, B)
    else:
        b[:] = a
    _, _, rc, rd = plan.getExchangeTables(2)
    temp = np.zeros((yo, zs, Nx), dtype=np.complex128)
    for p in range(P1):
        xsp, x0 = opl.in_block(p * P2 + j)[0][0], opl.in_block(p * P2 + j)[1][0]
        assert rc[p] == 16 * xsp * yo * zs
        temp[:, :, x0:x0 + xsp] = b[rd[p] // 16: rd[p] // 16 + xsp * yo * zs].reshape(yo, zs, xsp)
    x = orc.fft1d(temp.reshape(yo * zs, Nx), -1)
    got = x.T.reshape(Nx, yo, zs)
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=21)
    want = np.fft.fftn(g)[:, ost[1]:ost[1] + yo, ost[2]:ost[2] + zs]
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err < 1e-12, err
    nexch = (P1 > 1) + 2 * (P2 > 1)
    assert tc.calls == nexch, (tc.calls, nexch)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok err={err:.2e}")


if __name__ == "__main__":
    main()
