"""Two-hop relay of the group exchanges (include/dfft_c.h: dfft_comm_set_option "relay"; csrc/comm.hip relay_alltoallv) on
virtual ranks of one GPU: every message is cut into nranks parts, two travel directly and the others through the ranks outside
the pair; all partners travel together, as two grouped send/receive schedules per exchange (hop 1, hop 2).  The bytes must land exactly where the direct exchange puts them, so the
spectrum and the round trip are BIT-identical to the direct run -- on even and uneven splits, pipeline depths 1-4, groups of
2, 3 and 4, C2C and R2C.  The reference's counterpart is its per-peer overlap (src/pencil/mpicufft_pencil_opt1.cpp:1116-1275);
the multi-process form of the same layer runs over gloo in tests/test_cpu_multiprocess.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402

from test_gpu_parity import NPDT, TOL_FWD, TOL_RT, rel, run_distributed, run_distributed_real  # noqa: E402

CASES = [((128, 64, 32), 2, 4), ((66, 50, 38), 2, 4), ((64, 64, 64), 3, 2), ((48, 40, 56), 4, 2), ((40, 36, 30), 2, 3)]


@pytest.mark.parametrize("relay", [1, 2, 3])
@pytest.mark.parametrize("chunks", [1, 2, 3, 4])
@pytest.mark.parametrize("shape,P1,P2", CASES)
def test_relayed_exchange_is_bit_identical_to_the_direct_one(shape, P1, P2, chunks, relay):
    prec = "double"
    plans, ins, spec_d, backs_d = run_distributed(shape, P1, P2, prec, chunks=chunks)
    _, _, spec_r, backs_r = run_distributed(shape, P1, P2, prec, chunks=chunks, comm_options={"relay": relay})
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7).astype(NPDT[prec]).astype(np.complex128)
    want = orc.fft3d_c2c(g, -1)
    scale = np.max(np.abs(want))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        assert np.array_equal(spec_d[r], spec_r[r]), (r, "spectrum differs from the direct exchange")
        assert np.array_equal(backs_d[r], backs_r[r]), (r, "round trip differs from the direct exchange")
        assert np.max(np.abs(spec_r[r] - want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < TOL_FWD[prec]
        assert rel(backs_r[r] / float(np.prod(shape)), ins[r]) < TOL_RT[prec]


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("shape,P1,P2", [((32, 32, 64), 2, 4), ((24, 40, 50), 2, 2)])
def test_relayed_r2c_c2r(shape, P1, P2, prec):
    """the reference's own API (execR2C / execC2R) with the uneven Nz/2 + 1 split: 33 = 9 + 8 + 8 + 8"""
    plans, ins, spec_d, backs_d = run_distributed_real(shape, P1, P2, prec)
    _, _, spec_r, backs_r = run_distributed_real(shape, P1, P2, prec, comm_options={"relay": 3})
    for r in range(len(plans)):
        assert np.array_equal(spec_d[r], spec_r[r]) and np.array_equal(backs_d[r], backs_r[r])
        assert rel(backs_r[r] / float(np.prod(shape)), ins[r]) < TOL_RT[prec]


def test_relay_option_values():
    world = dfft.Comm.local(4)
    for v in (0, 1, 2, 3):
        world.setOption("relay", v)
    with pytest.raises(dfft.DfftError, match="relay"):
        world.setOption("relay", 4)


def test_relay_leaves_whole_world_groups_alone():
    """a slab plan's single exchange spans the whole world: there is nobody to relay through, the direct path runs"""
    shape = (64, 32, 16)
    plans, ins, spec_d, backs_d = run_distributed(shape, 4, 1, "double")
    _, _, spec_r, backs_r = run_distributed(shape, 4, 1, "double", comm_options={"relay": 3})
    for r in range(4):
        assert np.array_equal(spec_d[r], spec_r[r]) and np.array_equal(backs_d[r], backs_r[r])


@pytest.mark.parametrize("overlap", [0, 1])
def test_relay_is_two_grouped_operations_per_exchange_and_chunk(overlap):
    """all partners of a relayed exchange travel together: hop 1 and hop 2 are ONE point-to-point schedule each (dfft_comm_get_counter
    "list"), whatever the group size -- exchange 1 of a 2 x 4 grid has three partners per rank -- plus one table gather (an
    all-to-all-v) per exchange table at its first use.  relay_overlap = 1 runs hop 1 of a chunk on the relay's side stream under hop 2
    of the chunk before (double-buffered staging); both settings are bit-identical to the direct exchange."""
    shape, P1, P2, chunks = (66, 50, 38), 2, 4, 3
    plans, ins, spec_d, backs_d = run_distributed(shape, P1, P2, "double", chunks=chunks)
    plans_r, _, spec_r, backs_r = run_distributed(shape, P1, P2, "double", chunks=chunks, comm_options={"relay": 3, "relay_overlap": overlap})
    for r in range(P1 * P2):
        assert np.array_equal(spec_d[r], spec_r[r]) and np.array_equal(backs_d[r], backs_r[r])
    cnt = plans_r[0].comm.counters()      # one communicator, shared by the 8 virtual ranks
    per_rank = 2 * 2 * chunks             # (exchange 1 + exchange 2) x (forward + inverse) x chunks
    assert cnt["relayed"] == 8 * per_rank and cnt["list"] == 2 * cnt["relayed"], cnt
    # per table at its first use: one gather and one one-word agreement (every rank could set up its staging), both all-to-all-v
    assert cnt["relay_meta"] == 8 * per_rank and cnt["relay_agree"] == cnt["relay_meta"] and cnt["alltoallv"] == 2 * cnt["relay_meta"], cnt
    assert plans[0].comm.counters() == {"alltoallv": 8 * per_rank, "list": 0, "relayed": 0, "relay_meta": 0, "relay_agree": 0}
    # a second transform gathers nothing again
    from concurrent.futures import ThreadPoolExecutor
    outs = [torch.zeros(pl.getDomainSize() // 16, dtype=torch.complex128, device="cuda") for pl in plans_r]
    tin = [torch.from_numpy(x).cuda() for x in ins]
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda r: plans_r[r].execC2C(outs[r], tin[r], dfft.FORWARD), range(8)))
    torch.cuda.synchronize()
    cnt2 = plans_r[0].comm.counters()
    assert cnt2["relay_meta"] == cnt["relay_meta"] and cnt2["relay_agree"] == cnt["relay_agree"] and cnt2["list"] == cnt["list"] + 8 * 2 * 2 * chunks, (cnt, cnt2)
    for r, pl in enumerate(plans_r):
        s = pl.getOutSize()
        assert np.array_equal(outs[r][:s[0] * s[1] * s[2]].cpu().numpy().reshape(s), spec_d[r])


def test_relay_overlap_option_values():
    world = dfft.Comm.local(4)
    for v in (0, 1):
        world.setOption("relay_overlap", v)
    with pytest.raises(dfft.DfftError, match="relay_overlap"):
        world.setOption("relay_overlap", 2)
    with pytest.raises(dfft.DfftError, match="counter"):
        import ctypes as C
        from distributedfft_amd._lib import check, lib
        v = C.c_long(0)
        check(lib().dfft_comm_get_counter(world._h, b"nonsense", C.byref(v)))


def test_relay_repeated_runs_stay_bit_identical():
    """fresh plans again and again on the grid whose exchange tables differ in size from chunk to chunk (so that the staging buffers
    are re-allocated on the way): every run must equal the direct exchange bit for bit.  (Round 5: with staging from the virtual-memory
    API, unmapped and re-created under eight enqueueing host threads, 9-12 of 12 round trips differed; staging is plain hipMalloc.)"""
    shape, P1, P2, chunks = (66, 50, 38), 2, 4, 3
    _, _, spec_d, backs_d = run_distributed(shape, P1, P2, "double", chunks=chunks)
    for rep in range(6):
        _, _, spec_r, backs_r = run_distributed(shape, P1, P2, "double", chunks=chunks, comm_options={"relay": 3, "relay_overlap": rep % 2})
        for r in range(P1 * P2):
            assert np.array_equal(spec_d[r], spec_r[r]), (rep, r)
            assert np.array_equal(backs_d[r], backs_r[r]), (rep, r)


@pytest.mark.parametrize("overlap", [0, 1])
def test_a_rank_that_cannot_set_up_its_staging_makes_every_rank_send_directly(monkeypatch, overlap):
    """everything that can fail locally in a relayed exchange (staging, side stream, events) happens at the first use of an exchange
    table, followed by one word of agreement; a failure on ONE rank (injected) turns the table into a direct exchange on EVERY rank
    instead of leaving the others inside a collective (round-5 advice).  Same bytes, no relayed exchange counted."""
    shape, P1, P2, chunks = (66, 50, 38), 2, 4, 3
    _, _, spec_d, backs_d = run_distributed(shape, P1, P2, "double", chunks=chunks)
    monkeypatch.setenv("DFFT_RELAY_TEST_FAIL_RANK", "5")
    plans_r, _, spec_r, backs_r = run_distributed(shape, P1, P2, "double", chunks=chunks, comm_options={"relay": 3, "relay_overlap": overlap})
    for r in range(P1 * P2):
        assert np.array_equal(spec_d[r], spec_r[r]) and np.array_equal(backs_d[r], backs_r[r])
    cnt = plans_r[0].comm.counters()
    assert cnt["relayed"] == 0 and cnt["list"] == 0 and cnt["relay_agree"] == cnt["relay_meta"] > 0, cnt
