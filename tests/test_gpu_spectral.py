"""Option "spectral_layout" = 1 (include/dfft_c.h): the spectrum block stays x-contiguous, [yo][zs][Nx], so that neither x pass
touches the point-major layout.  Sizes, starts and exchange tables are the reference's (include/mpicufft_pencil.hpp:94-122); only
the order inside `out` differs, and dfft_get_out_strides / spectrumView index it.  Checked against the oracle at every point,
through the round trip and through the reference's testcase 4 (forward -> pointwise work on the spectrum -> inverse,
tests/src/pencil/random_dist_3D.cu:685-811), on one rank and on virtual ranks, C2C and R2C, fp64 and fp32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402

from test_gpu_parity import NPDT, TOL_FWD, TOL_RT, rel, run_distributed, run_distributed_real  # noqa: E402

OPT = {"spectral_layout": 1}
GRIDS = [((32, 16, 64), 1, 1), ((64, 64, 64), 2, 2), ((128, 64, 32), 2, 4), ((66, 50, 38), 2, 3), ((64, 32, 16), 4, 1), ((16, 32, 48), 1, 4),
         ((1024, 16, 8), 2, 2), ((2048, 8, 16), 2, 1)]


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("shape,P1,P2", GRIDS)
def test_x_contiguous_spectrum_c2c(shape, P1, P2, prec):
    plans, ins, spec, backs = run_distributed(shape, P1, P2, prec, options=OPT)
    ref_plans, _, spec_ref, _ = run_distributed(shape, P1, P2, prec)
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7).astype(NPDT[prec]).astype(np.complex128)
    want = orc.fft3d_c2c(g, -1)
    scale = np.max(np.abs(want))
    for r, pl in enumerate(plans):
        s, o, st = pl.getOutSize(), pl.getOutStart(), pl.getOutStrides()
        assert (s, o) == (ref_plans[r].getOutSize(), ref_plans[r].getOutStart()) and st == (1, s[2] * s[0], s[0])
        assert np.max(np.abs(spec[r] - want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < TOL_FWD[prec]
        # the same values as the reference layout holds (not bit for bit: the x pass may run another radix chain for its natural-line store)
        assert np.max(np.abs(spec[r] - spec_ref[r])) / scale < TOL_FWD[prec]
        assert rel(backs[r] / float(np.prod(shape)), ins[r]) < TOL_RT[prec]


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("shape,P1,P2", [((16, 16, 32), 1, 1), ((32, 32, 64), 2, 4), ((24, 40, 50), 2, 2), ((64, 32, 16), 4, 1)])
def test_x_contiguous_spectrum_r2c(shape, P1, P2, prec):
    """the reference's own API: execR2C leaves the Hermitian half (Nz/2 + 1 split over P2) x-contiguous, execC2R takes it back"""
    plans, ins, spec, backs = run_distributed_real(shape, P1, P2, prec, options=OPT)
    g = orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13).astype(np.float32 if prec == "float" else np.float64).astype(np.float64)
    want = np.fft.rfftn(g)
    scale = np.max(np.abs(want))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        assert np.max(np.abs(spec[r] - want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < TOL_FWD[prec]
        assert rel(backs[r] / float(np.prod(shape)), ins[r]) < TOL_RT[prec]


@pytest.mark.parametrize("shape,P1,P2", [((32, 32, 32), 1, 1), ((32, 32, 32), 2, 4), ((64, 32, 16), 2, 2)])
def test_testcase4_laplacian_through_the_strides(shape, P1, P2):
    """reference testcase 4 (random_dist_3D.cu:685-811): u = sin sin sin, the spectrum times -(k1^2+k2^2+k3^2)/sqrt(N^3) through the
    (Nx, yo, zs) view of the x-contiguous block, inverse, compare with -3 sqrt(N^3) u"""
    Nx, Ny, Nz = shape
    x, y, z = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz), indexing="ij")
    u = np.sin(2 * np.pi * x / Nx) * np.sin(2 * np.pi * y / Ny) * np.sin(2 * np.pi * z / Nz)

    def modify(blk, s, o):
        orc.derivative_coefficients(blk, shape, o[2], o[1], half=True)

    plans, ins, spec, backs = run_distributed_real(shape, P1, P2, "double", field=u, modify=modify, options=OPT)
    n3 = float(Nx * Ny * Nz)
    for r in range(len(plans)):
        assert np.max(np.abs(backs[r] - orc.testcase4_expected(shape, ins[r]))) < 1e-9 * np.sqrt(n3)


def test_tuner_and_pipeline_depths_with_the_x_contiguous_spectrum():
    shape, P1, P2 = (64, 48, 40), 2, 2
    for chunks in (1, 3):
        plans, ins, spec, backs = run_distributed(shape, P1, P2, "double", chunks=chunks, options=OPT)
        _, _, spec_ref, _ = run_distributed(shape, P1, P2, "double", chunks=chunks)
        for r in range(4):
            assert np.max(np.abs(spec[r] - spec_ref[r])) / np.max(np.abs(spec_ref[r])) < 1e-11 and rel(backs[r] / float(np.prod(shape)), ins[r]) < 1e-10
    # one rank: dfft_tune_variants on a plan whose inverse runs the mirrored order
    n = 64
    pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), None, precision="double", rank=0)
    pl.setOption("spectral_layout", 1)
    pl.initFFT(dfft.GlobalSize(n, n, n), dfft.Partition(1, 1), True, c2c=True)
    g = orc.fill_block((n, n, n), (0, 0, 0), (n, n, n), 2, seed=3)
    d_in = torch.from_numpy(g).cuda()
    d_out = torch.zeros(pl.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    d_back = torch.zeros_like(d_in)
    trials = pl.tuneVariants(d_in, d_out, d_back)
    assert len(trials) >= 2
    pl.execC2C(d_out, d_in, dfft.FORWARD)
    got = pl.spectrumView(d_out).contiguous().cpu().numpy()
    want = orc.fft3d_c2c(g, -1)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-11
    pl.execC2C(d_back, d_out, dfft.INVERSE)
    assert rel(d_back.cpu().numpy() / n ** 3, g) < 1e-10
