"""What one GPU can exercise of the native RCCL transport (csrc/comm.hip RcclComm) before a multi-GPU node ever sees it: a world of
one rank whose own block travels through ncclSend / ncclRecv to itself ("self_send"), on every channel's communicator
("dup_channel" = 3: three ncclCommSplit duplicates), between virtual-memory ranges from the library's allocator, as an all-to-all-v
and as the point-to-point schedule the relay's hops are made of (several pieces to ONE peer in one group, matched in order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def rccl_world_of_one():
    comm = dfft.Comm.rccl(dfft.Comm.rccl_unique_id(), 1, 0)
    comm.setOption("self_send", 1)
    comm.setOption("dup_channel", 3)
    yield comm
    comm.setOption("dup_channel", 0)
    comm.destroy()


def test_every_channel_carries_an_alltoallv_between_library_buffers(rccl_world_of_one):
    comm = rccl_world_of_one
    assert comm.info() == (1, 1)      # ncclCommCount of the communicator itself
    n = 64 << 20
    a, b = dfft.DeviceBuffer.alloc(n), dfft.DeviceBuffer.alloc(n)      # virtual-memory ranges (1 GiB-chunk recipe, 64 MiB: one chunk)
    ta, tb = a.tensor(torch.int64), b.tensor(torch.int64)
    for ch in range(4):
        comm.setOption("test_channel", ch)
        ta.copy_(torch.arange(n // 8, device="cuda") * (ch + 3))
        tb.zero_()
        torch.cuda.synchronize()
        comm.alltoallv(0, a, [n - 4096], [4096], b, [n - 4096], [0], [0], 0)
        torch.cuda.synchronize()
        assert torch.equal(tb[:(n - 4096) // 8], ta[512:]), ch
    comm.setOption("test_channel", 0)
    cnt = comm.counters()
    assert cnt["alltoallv"] >= 4 and cnt["list"] == 0
    del ta, tb
    a.free(); b.free()


def test_point_to_point_schedule_with_several_pieces_per_peer(rccl_world_of_one):
    """one ncclGroup with three sends and three receives between the same pair of ranks: RCCL must match them in the order they were
    issued (the relay lists a link's pieces by layer at both ends)"""
    comm = rccl_world_of_one
    n = 32 << 20
    a, b = dfft.DeviceBuffer.alloc(n), dfft.DeviceBuffer.alloc(n)
    ta, tb = a.tensor(torch.uint8), b.tensor(torch.uint8)
    ta.copy_(torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda"))
    tb.zero_()
    pieces = [(0, 1 << 20, 0), (3 << 20, 5 << 20, 2 << 20), (10 << 20, (2 << 20) + 256, 9 << 20)]      # (send offset, bytes, receive offset)
    for ch in (0, 2, 3):
        comm.setOption("test_channel", ch)
        tb.zero_()
        torch.cuda.synchronize()
        before = comm.counters()["list"]
        # listed out of order on purpose: the layer, not the position, pairs a send with its receive
        sends = [(0, layer, a.address + so, nb) for layer, (so, nb, ro) in enumerate(pieces)][::-1]
        recvs = [(0, layer, b.address + ro, nb) for layer, (so, nb, ro) in enumerate(pieces)]
        comm.sendrecvList(0, sends, recvs, len(pieces))
        torch.cuda.synchronize()
        assert comm.counters()["list"] == before + 1
        for so, nb, ro in pieces:
            assert torch.equal(tb[ro:ro + nb], ta[so:so + nb]), (ch, so)
    comm.setOption("test_channel", 0)
    del ta, tb
    a.free(); b.free()


def test_plan_on_the_rccl_world_of_one(rccl_world_of_one):
    """a plan whose communicator is the RCCL one (no exchange partner: the passes run, the comm streams are created) at every point"""
    shape = (48, 40, 64)
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=9)
    want = orc.fft3d_c2c(g, -1)
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), rccl_world_of_one, precision="double")
    plan.setPipelineChunks(4)
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), True, c2c=True)
    x = torch.from_numpy(g).cuda()
    out = dfft.DeviceBuffer.alloc(plan.getDomainSize())
    back = torch.zeros_like(x)
    plan.execC2C(out, x, dfft.FORWARD)
    got = out.tensor(torch.complex128)[:g.size].cpu().numpy().reshape(shape)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-11
    plan.execC2C(back, out, dfft.INVERSE)
    assert float((back / g.size - x).abs().max()) / 255.0 < 1e-10
    del plan
    out.free()
