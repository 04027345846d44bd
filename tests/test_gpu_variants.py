"""Every kernel configuration (role variant of csrc/cfg_f64.hip.h / cfg_f32.hip.h) on EVERY pass: dfft_tune_variants sets all passes of
a plan to one configuration number at a time and keeps, per pass, what runs faster -- so each configuration must be a correct
transform under every load / store address form (natural lines, tiled segments of several peers and chunks, transposed tiles,
point-major with even and odd row pitch), not only in the role it was written for.  Forced here through the variant_* options on
one rank and on 2 x 2 virtual ranks, C2C and R2C, against the oracle; lengths without the configuration fall back to their default."""
import functools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402,F401
from oracle import oracle as orc  # noqa: E402
from test_gpu_parity import NPDT, rel, run_distributed, run_distributed_real  # noqa: E402

VARIANTS = {"double": [1, 2, 3, 7, 8], "float": [1, 3, 4, 5, 6, 7, 9, 14, 15]}
PASSES = ("fz", "fy", "fx", "ix", "iy", "iz")
TF = {"double": 1e-11, "float": 1e-4}
TR = {"double": 1e-10, "float": 5e-5}


@functools.lru_cache(maxsize=4)
def spectrum(shape, prec, real):
    if real:
        rdt = np.float64 if prec == "double" else np.float32
        return orc.fft3d_r2c(orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13).astype(rdt).astype(np.float64))
    return orc.fft3d_c2c(orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7).astype(NPDT[prec]).astype(np.complex128), -1)


def check(shape, P1, P2, prec, v, real):
    opts = {"variant_" + k: v for k in PASSES}
    run = run_distributed_real if real else run_distributed
    plans, ins, spec, backs = run(shape, P1, P2, prec, options=opts)
    want = spectrum(shape, prec, real)
    n3 = float(np.prod(shape))
    for r, pl in enumerate(plans):
        s, o = pl.getOutSize(), pl.getOutStart()
        ref = want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(spec[r] - ref)) / np.max(np.abs(want)) < TF[prec], (shape, P1, P2, v)
        assert rel(backs[r] / n3, ins[r]) < TR[prec], (shape, P1, P2, v)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_every_configuration_on_the_y_and_x_passes(N, prec):
    """grid N x N x 16: tiled / transposed-tile / point-major forms on N-point lines; one rank (both pass orders share the
    forward launches) and 2 x 2 ranks with two pipeline chunks (segmented sides)"""
    for v in VARIANTS[prec]:
        check((N, N, 16), 1, 1, prec, v, False)
        check((N, N, 16), 2, 2, prec, v, False)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("N", [512, 1024, 2048])
def test_every_configuration_on_the_z_passes(N, prec):
    """grid 16 x 24 x N: natural lines on one side of N-point lines"""
    for v in VARIANTS[prec]:
        check((16, 24, N), 1, 1, prec, v, False)
        check((16, 24, N), 2, 2, prec, v, False)


@pytest.mark.parametrize("prec", ["double", "float"])
def test_every_configuration_on_odd_pitch_rows(prec):
    """R2C plan, 17-wide spectrum rows (1024 x 512 x 32): the x pass stores point-major rows of odd pitch (shifted tile windows at
    fp64), the inverse x pass reads them"""
    for v in VARIANTS[prec]:
        check((1024, 512, 32), 1, 1, prec, v, True)
        check((1024, 512, 32), 2, 2, prec, v, True)
