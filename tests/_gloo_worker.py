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


def sequence_main(seq, rank, world, shape):
    """Z_Then_YX / Y_Then_ZX with the reference's opt0 buffer layouts (src/slab/z_then_yx/
    mpicufft_slab_z_then_yx.cpp:190-196, src/slab/y_then_zx/mpicufft_slab_y_then_zx.cpp:309-319):
    oracle 1-D FFTs, the product's plan tables and exchange over gloo"""
    P = world
    tc = TorchComm(dist, rank, world, P, 1)
    cls = dfft.MPIcuFFT_Slab_Z_Then_YX if seq == "zyx" else dfft.MPIcuFFT_Slab_Y_Then_ZX
    plan = cls(dfft.Configurations(), tc, precision="double", rank=rank)
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Slab_Partition(P), allocate=False, c2c=True)
    Nx, Ny, Nz = shape
    isz, ist, osz, ost = plan.getInSize(), plan.getInStart(), plan.getOutSize(), plan.getOutStart()
    xs = isz[0]
    nel = plan.getDomainSize() // 16
    A, B = torch.zeros(nel, dtype=torch.complex128), torch.zeros(nel, dtype=torch.complex128)
    tc.register(A)
    tc.register(B)
    a, b = A.numpy(), B.numpy()
    blk = orc.fill_block(shape, ist, isz, 2, seed=21)
    sc, sd, rc, rd = plan.getExchangeTables(2)
    if seq == "zyx":
        first = orc.fft1d(blk.reshape(xs * Ny, Nz), -1).reshape(xs, Ny, Nz)                 # z pass
        cut = [sum(v // (16 * Ny * xs) for v in sc[:q]) for q in range(P + 1)]              # z split
        for q in range(P):      # pack [xs][Ny][zs[q]]
            a[sd[q] // 16: (sd[q] + sc[q]) // 16] = first[:, :, cut[q]:cut[q + 1]].ravel()
    else:
        first = orc.fft1d(np.ascontiguousarray(blk.transpose(0, 2, 1)).reshape(xs * Nz, Ny), -1).reshape(xs, Nz, Ny).transpose(0, 2, 1)   # y pass
        cut = [sum(v // (16 * Nz * xs) for v in sc[:q]) for q in range(P + 1)]              # y split
        for q in range(P):      # pack [xs][yo[q]][Nz]
            a[sd[q] // 16: (sd[q] + sc[q]) // 16] = first[:, cut[q]:cut[q + 1], :].ravel()
    send_copy = a.copy()
    plan.exchange(2, dfft.FORWARD, A, B)
    # the received blocks stack along x: [Nx][Ny][zs] resp. [Nx][yo][Nz]
    mid = np.zeros(osz, dtype=np.complex128)
    x0 = 0
    for q in range(P):
        n = rc[q] // 16
        xq = n // (osz[1] * osz[2])
        mid[x0:x0 + xq] = b[rd[q] // 16: rd[q] // 16 + n].reshape(xq, osz[1], osz[2])
        x0 += xq
    assert x0 == Nx
    A.zero_()
    plan.exchange(2, dfft.INVERSE, B, A)
    assert np.array_equal(a, send_copy), "the inverse exchange is not the mirror"
    other = 1 if seq == "zyx" else 2       # remaining axes: (y, x) resp. (z, x)
    t = np.moveaxis(mid, other, -1)
    t = orc.fft1d(np.ascontiguousarray(t).reshape(-1, shape[other]), -1).reshape(t.shape)
    mid = np.moveaxis(t, -1, other)
    t = np.moveaxis(mid, 0, -1)
    t = orc.fft1d(np.ascontiguousarray(t).reshape(-1, Nx), -1).reshape(t.shape)
    got = np.moveaxis(t, -1, 0)
    want = np.fft.fftn(orc.fill_block(shape, (0, 0, 0), shape, 2, seed=21))
    want = want[:, ost[1]:ost[1] + osz[1], ost[2]:ost[2] + osz[2]]
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err < 1e-12, err
    assert tc.calls == 2
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok err={err:.2e}")


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    P1, P2 = int(sys.argv[1]), int(sys.argv[2])
    shape = tuple(int(v) for v in sys.argv[3].split("x"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if len(sys.argv) > 4:
        return sequence_main(sys.argv[4], rank, world, shape)
    layered = os.environ.get("DFFT_TEST_NO_LIST") == "1"      # no schedule callback: a relay hop runs as group - 1 all-to-all-v layers
    tc = TorchComm(dist, rank, world, P1, P2, list_callback=not layered)
    relay = int(os.environ.get("DFFT_TEST_RELAY", "0"))
    if relay:       # two-hop relay of the group exchanges (include/dfft_c.h: dfft_comm_set_option "relay")
        tc.setOption("relay", relay)
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
    # a relayed exchange (group = a strict subset of the world) is one table gather at its first use (an all-to-all-v) plus TWO
    # point-to-point schedules -- hop 1 and hop 2, all partners together -- whatever the group size
    def ncalls(bit, ng):
        if ng <= 1:
            return 0, 0
        if relay & bit and ng < world:
            return 1, 2
        return 1, 0
    c2, l2 = ncalls(1, P1)
    c1, l1 = ncalls(2, P2)
    cnt = tc.comm.counters()
    nrel = (1 if l2 else 0) + (2 if l1 else 0)
    assert cnt["relayed"] == nrel and cnt["relay_meta"] == nrel, cnt
    # ... and one one-word agreement per relayed table at its first use (every rank could set up its staging: comm.hip relay_agree)
    agree = cnt["relay_agree"]
    assert agree == nrel, cnt
    if not layered:
        assert tc.calls == c2 + 2 * c1 + agree and tc.p2p_calls == 0 and tc.list_calls == l2 + 2 * l1, (tc.calls, tc.p2p_calls, tc.list_calls, c2, c1, l2, l1)
        assert cnt["list"] == 2 * nrel and cnt["alltoallv"] == tc.calls, (cnt, tc.calls)
    else:
        # every hop of a relayed exchange is group - 1 all-to-all-v layers whose pieces are not back to back (the transport's per-peer path)
        layers = (2 * (P1 - 1) if l2 else 0) + (2 * 2 * (P2 - 1) if l1 else 0)
        assert cnt["list"] == 0 and tc.list_calls == 0 and cnt["alltoallv"] == tc.calls == c2 + 2 * c1 + agree + layers, (cnt, tc.calls, layers)
        assert tc.p2p_calls == layers, (tc.p2p_calls, layers)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok err={err:.2e}")


if __name__ == "__main__":
    main()
