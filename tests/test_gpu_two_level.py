"""Two-level lines (N = N1*N2 over two launches of the generic kernel: csrc/fft_pass.hip.h "Two-level lines", dfft.hip
axis_plan_two_level): the lengths that have no kernel of their own -- beyond 8192, beyond 4096 when not a power of two --
which the reference takes through cuFFT (mpicufft_pencil_opt1.cpp:165-197: cufftMakePlanMany64 accepts any size).

* kernel level: natural lines, both directions, both precisions, forced on short lengths (variant -2: every pairing of
  plain and Bluestein levels) and by itself on long ones;
* plans: long axes in every position, C2C and R2C (the real modes of the two levels), on one rank and distributed;
* option two_level = 1 forces the two-level form on small grids for every pipeline (pencil, slab, the two slab
  sequences, chunked exchanges, partial transforms), compared with the oracle and with the one-launch kernels.

Tolerances as in test_gpu_parity.py (fp64 1e-11 forward / 1e-10 round trip, fp32 1e-4 / 5e-5), relaxed by sqrt-ish growth on
the lines of 10^4 .. 10^5 points where stated."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_gpu_parity import NPDT, rel, run_distributed, run_distributed_real  # noqa: E402
import test_gpu_slab_sequences as slabs  # noqa: E402

FORCED = [4, 6, 8, 9, 15, 16, 21, 64, 100, 121, 256, 1000, 1024, 1155, 2048, 3000, 4096, 6 * 1024, 8192]
LONG = [4098, 5000, 6000, 9999, 10000, 12288, 16384, 20000, 65536, 3 * 4093, 100000, 131072]


def fft1d_case(N, prec, variant, batch, big_prime=False):
    rng = np.random.default_rng(N)
    x = (rng.uniform(0, 255, (batch, N)) + 1j * rng.uniform(0, 255, (batch, N))).astype(NPDT[prec])
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros_like(d_in)
    grow = max(1.0, np.log2(N) / 12.0)
    for direction in (dfft.FORWARD, dfft.INVERSE):
        torch.cuda.synchronize()
        dfft.fft1d_batched(d_out, d_in, N, batch, direction, prec, variant=variant)
        torch.cuda.synchronize()
        ref = np.fft.fft(x.astype(np.complex128), axis=-1) if direction == dfft.FORWARD else np.fft.ifft(x.astype(np.complex128), axis=-1) * N
        if big_prime:      # the oracle transforms a prime length as an O(N^2) sum (65537 points: 25 s per case): pocketfft alone here
            want = ref
        else:
            want = orc.fft1d(x.astype(np.complex128), direction)
            assert rel(want, ref) < 1e-12      # independent cross-check of the oracle itself on these lengths
        assert rel(d_out.cpu().numpy(), want) < (2e-11 if prec == "double" else 2e-4) * grow
    assert np.array_equal(d_in.cpu().numpy(), x), "the pass must not modify its input"


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("N", FORCED)
def test_fft1d_two_level_forced_vs_oracle(N, prec):
    """variant -2: two levels wherever the length splits -- plain x plain (64, 1024), plain x Bluestein (6, 3000),
    Bluestein x Bluestein (15, 121, 1155), sub-tile inner transforms (8192 = 4096 x 2 is avoided by the cost model, 6144 not)"""
    fft1d_case(N, prec, -2, 45 if N < 4096 else 19)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("N", LONG)
def test_fft1d_long_lines_vs_oracle(N, prec):
    """lengths without a kernel of their own take the two-level form by themselves (ragged batch: not a multiple of the tile)"""
    fft1d_case(N, prec, 0, 11 if N < 50000 else 5)


LONG_BLUESTEIN = [4099, 8198, 5003, 10007, 12289, 2 * 3 * 4099, 65537]


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("N", LONG_BLUESTEIN)
def test_fft1d_long_bluestein_lines_vs_oracle(N, prec):
    """lengths that neither fit one launch nor split into two factors within its reach -- a prime above 4096 (4099, 5003, 10007,
    12289, 65537), a small multiple of one (8198, 24594) -- run Bluestein's algorithm with M-point transforms that are two-level
    lines themselves (four launches of the generic kernel, dfft.hip launch_long_bluestein).  The reference takes such sizes through
    cuFFT like any other (mpicufft_pencil_opt1.cpp:165-197)."""
    assert dfft.axis_plan_info(N, prec)["kind"] == "long_bluestein"
    fft1d_case(N, prec, 0, 7 if N < 20000 else 3, big_prime=N > 20000)


def test_every_length_has_a_plan_and_absurd_ones_fail_loudly():
    """every length from 2 to 2^23 has a plan now; beyond the 32-bit point indices of the kernel (2^24) there is none"""
    for N in (2, 3, 4099, 8191 * 3, 99991, 1000003, (1 << 23) - 15):
        assert dfft.axis_plan_info(N, "double") is not None
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    with pytest.raises(dfft.DfftError, match="unsupported axis length"):
        plan.initFFT(dfft.GlobalSize((1 << 24) + 1, 4, 4), dfft.Pencil_Partition(1, 1), False, c2c=True)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape", [(4099, 4, 6), (3, 4099, 8), (5, 4, 4099), (6, 5, 8198), (8198, 3, 10)])
def test_single_rank_long_bluestein_axes_vs_oracle(shape, c2c, prec):
    """a long-Bluestein axis in every position; R2C: real input lines of an odd (4099) and an even (8198) prime-ridden length"""
    g, got, back = single(shape, prec, c2c)
    want = orc.fft3d_c2c(g.astype(np.complex128), -1) if c2c else orc.fft3d_r2c(g.astype(np.float64))
    assert rel(got, want) < (4e-11 if prec == "double" else 4e-4)
    assert rel(back / g.size, g) < (2e-10 if prec == "double" else 1e-4)


@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P1,P2,chunks", [((4099, 8, 12), 2, 2, 2), ((8, 4099, 6), 2, 2, 1), ((6, 8, 4099), 2, 3, 2), ((4099, 6, 8), 3, 1, 3)])
def test_distributed_long_bluestein_axes_vs_oracle(shape, P1, P2, chunks, c2c):
    """... distributed: the first and the last launch carry the pass's own (segmented, chunked) address forms"""
    prec = "double"
    if c2c:
        plans, ins, spec, backs = run_distributed(shape, P1, P2, prec, chunks=chunks)
        g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7).astype(np.complex128)
        want = orc.fft3d_c2c(g, -1)
    else:
        plans, ins, spec, backs = run_distributed_real(shape, P1, P2, prec)
        g = orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13).astype(np.float64)
        want = orc.fft3d_r2c(g)
    scale = np.max(np.abs(want))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        assert np.max(np.abs(spec[r] - want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < 4e-11
        assert rel(backs[r] / float(np.prod(shape)), ins[r]) < 2e-10


def single(shape, prec, c2c, options=None, seed=5):
    cdt = torch.complex128 if prec == "double" else torch.complex64
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision=prec)
    for k, v in (options or {}).items():
        plan.setOption(k, v)
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), True, c2c=c2c)
    esz = 16 if prec == "double" else 8
    g = orc.fill_block(shape, (0, 0, 0), shape, 2 if c2c else 1, seed=seed)
    g = g.astype(NPDT[prec]) if c2c else g.astype(np.float64 if prec == "double" else np.float32)
    d_in = torch.from_numpy(g).cuda()
    d_out = torch.zeros(plan.getDomainSize() // esz, dtype=cdt, device="cuda")
    d_back = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    if c2c:
        plan.execC2C(d_out, d_in, dfft.FORWARD)
    else:
        plan.execR2C(d_out, d_in)
    s = plan.getOutSize()
    got = d_out[:s[0] * s[1] * s[2]].cpu().numpy().reshape(s)
    assert np.array_equal(d_in.cpu().numpy(), g), "forward must not modify its input"
    torch.cuda.synchronize()
    if c2c:
        plan.execC2C(d_back, d_out, dfft.INVERSE)
    else:
        plan.execC2R(d_back, d_out)
    torch.cuda.synchronize()
    return g, got, d_back.cpu().numpy()


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape", [(16384, 4, 6), (3, 10000, 8), (5, 4, 16384), (9, 3, 10000), (2, 3, 9999), (12288, 5, 20),
                                   (4100, 2, 5000)])
def test_single_rank_long_axes_vs_oracle(shape, c2c, prec):
    """a long axis in every position; R2C: the real modes of the two levels (even and odd Nz)"""
    g, got, back = single(shape, prec, c2c)
    want = orc.fft3d_c2c(g.astype(np.complex128), -1) if c2c else orc.fft3d_r2c(g.astype(np.float64))
    assert rel(got, want) < (4e-11 if prec == "double" else 4e-4)
    assert rel(back / g.size, g) < (2e-10 if prec == "double" else 1e-4)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape", [(8, 8, 8), (12, 10, 14), (64, 64, 64), (30, 16, 50), (9, 15, 21), (128, 4, 36), (16, 256, 10)])
def test_single_rank_forced_two_level_vs_oracle(shape, c2c, prec):
    """option two_level = 1: every axis that splits runs in two levels (primes keep Bluestein), on both pass orders of a
    single rank; equal to the oracle and, to rounding, to the plan without the option"""
    g, got, back = single(shape, prec, c2c, {"two_level": 1})
    want = orc.fft3d_c2c(g.astype(np.complex128), -1) if c2c else orc.fft3d_r2c(g.astype(np.float64))
    assert rel(got, want) < (2e-11 if prec == "double" else 2e-4)
    assert rel(back / g.size, g) < (1e-10 if prec == "double" else 5e-5)
    _, plain, _ = single(shape, prec, c2c)
    assert rel(got, plain) < (2e-11 if prec == "double" else 2e-4)
    if c2c:
        g2, got2, back2 = single(shape, prec, c2c, {"two_level": 1, "single_order": 1})
        assert rel(got2, want) < (2e-11 if prec == "double" else 2e-4)
        assert rel(back2 / g.size, g) < (1e-10 if prec == "double" else 5e-5)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("shape,P1,P2,chunks", [((16, 16, 16), 2, 2, None), ((32, 32, 32), 2, 4, 2), ((12, 10, 14), 2, 4, None),
                                                ((64, 32, 16), 4, 2, 4), ((30, 20, 18), 3, 1, None), ((16, 32, 16), 1, 4, 1),
                                                ((64, 64, 64), 3, 5, None)])
def test_distributed_forced_two_level_vs_oracle(shape, P1, P2, chunks, prec):
    """pencil and slab pipelines with segmented (multi-peer, chunked) loads and stores on two-level axes, C2C and R2C"""
    opts = {"two_level": 1}
    plans, ins, spec, backs = run_distributed(shape, P1, P2, prec, chunks=chunks, options=opts)
    want = orc.fft3d_c2c(orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7).astype(NPDT[prec]).astype(np.complex128), -1)
    n3 = float(np.prod(shape))
    tf, tr = (2e-11, 1e-10) if prec == "double" else (2e-4, 5e-5)
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        ref = want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(spec[r] - ref)) / np.max(np.abs(want)) < tf
        assert rel(backs[r] / n3, ins[r]) < tr
    plans, ins, spec, backs = run_distributed_real(shape, P1, P2, prec, options=opts)
    rdt = np.float64 if prec == "double" else np.float32
    wantr = orc.fft3d_r2c(orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13).astype(rdt).astype(np.float64))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        ref = wantr[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(spec[r] - ref)) / np.max(np.abs(wantr)) < tf
        assert rel(backs[r] / n3, ins[r]) < tr


@pytest.mark.parametrize("shape,P1,P2", [((8200, 8, 6), 2, 2), ((6, 9000, 10), 2, 2), ((4, 6, 10000), 2, 2), ((16384, 4, 8), 4, 1)])
def test_distributed_long_axes_vs_oracle(shape, P1, P2):
    """long axes under a partition (uneven splits included), C2C and R2C, fp64"""
    plans, ins, spec, backs = run_distributed(shape, P1, P2, "double")
    want = orc.fft3d_c2c(orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7), -1)
    n3 = float(np.prod(shape))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        ref = want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(spec[r] - ref)) / np.max(np.abs(want)) < 4e-11
        assert rel(backs[r] / n3, ins[r]) < 2e-10
    plans, ins, spec, backs = run_distributed_real(shape, P1, P2, "double")
    wantr = orc.fft3d_r2c(orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        ref = wantr[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(spec[r] - ref)) / np.max(np.abs(wantr)) < 4e-11
        assert rel(backs[r] / n3, ins[r]) < 2e-10


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P", [((16, 16, 16), 2), ((33, 20, 18), 4), ((24, 10, 20), 4), ((64, 32, 16), 8)])
def test_slab_sequences_forced_two_level(shape, P, c2c, prec):
    """Z_Then_YX (forward and inverse) and Y_Then_ZX (forward; R2C along y: real lines at a stride) on two-level axes"""
    tf, tr = (2e-11, 1e-10) if prec == "double" else (2e-4, 5e-5)
    plans, ins, spec, backs = slabs.run(dfft.MPIcuFFT_Slab_Z_Then_YX, shape, P, prec, c2c, options={"two_level": 1})
    g = slabs.global_input(shape, c2c, prec)
    want = orc.fft3d_c2c(g, -1) if c2c else orc.fft3d_r2c(g)
    n3 = float(np.prod(shape))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        assert np.max(np.abs(spec[r] - want[:, :, o[2]:o[2] + s[2]])) / np.max(np.abs(want)) < tf
        assert np.max(np.abs(backs[r] / n3 - ins[r])) / 255.0 < tr
    plans, spec = slabs.run_yzx(shape, P, prec, c2c, options={"two_level": 1})
    g = slabs.global_input(shape, c2c, prec, seed=33)
    want = orc.fft3d_c2c(np.ascontiguousarray(g.astype(np.complex128)), -1)
    want = want[:, :(shape[1] if c2c else shape[1] // 2 + 1), :]
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        assert np.max(np.abs(spec[r] - want[:, o[1]:o[1] + s[1], :])) / np.max(np.abs(want)) < tf


def test_work_area_holds_the_level_scratch():
    """getWorkSizeDevice grows by the scratch between the levels, and a caller-owned work area of that size is enough"""
    shape = (10000, 6, 8)
    sizes = {}
    for name, shp in (("long", shape), ("short", (1000, 6, 8))):
        plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
        plan.initFFT(dfft.GlobalSize(*shp), dfft.Pencil_Partition(1, 1), False, c2c=True)
        sizes[name] = (plan.getWorkSizeDevice(), plan.getDomainSize())
    assert sizes["short"][0] <= sizes["short"][1] + 256
    assert sizes["long"][0] >= 2 * sizes["long"][1]      # one slice + tiles x 8 lines x 10000 points of scratch
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), False, c2c=True)
    work = torch.zeros(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
    plan.setWorkArea(work)
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=3)
    d_in = torch.from_numpy(g).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    plan.execC2C(d_out, d_in, dfft.FORWARD)
    torch.cuda.synchronize()
    assert rel(d_out.cpu().numpy().reshape(shape), orc.fft3d_c2c(g, -1)) < 4e-11
