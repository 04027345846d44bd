"""CPU-side checks: the C ABI library loads and exports every symbol include/dfft_c.h declares,
the host-side plan algebra equals the oracle restatement of the reference's formulas, and the
product has no CPU compute path."""
import ctypes
import os
import re

import pytest

import distributedfft_amd as dfft
from distributedfft_amd import _lib
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dfft_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfft_[a-z0-9_]+)\s*\(", src)) - {"dfft_alltoallv_fn"})


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dfft_c.h but not exported"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert bound == set(names), (bound ^ set(names))
    _lib.lib()
    assert b"gfx950" in _lib.lib().dfft_version()


def test_kernel_table():
    for prec in ("double", "float"):
        for n in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
            info = dfft.kernel_info(n, prec)
            assert info and info["threads"] <= 1024 and info["lds_bytes"] <= 160 * 1024
            assert info["lines_per_workgroup"] % (8 if prec == "double" else 16) == 0
        # other lengths run through Bluestein on the next power of two >= 2N-1 (N <= 1024)
        assert dfft.kernel_info(3, prec)["points_per_thread"] == 8 and dfft.kernel_info(1000, prec)["threads"] > 0
        assert dfft.kernel_info(4096, prec) is None and dfft.kernel_info(1025, prec) is None


CASES = [((12, 10, 14), 2, 4, False), ((9, 7, 10), 3, 2, True), ((1000, 100, 30), 4, 2, False), ((1024, 1024, 1024), 2, 4, False), ((1024, 1024, 1024), 2, 4, True), ((512, 512, 512), 2, 1, True),
         ((64, 32, 16), 4, 2, False), ((16, 16, 16), 3, 2, True), ((2048, 2048, 2048), 2, 4, True),
         ((256, 256, 256), 1, 1, True), ((64, 64, 64), 3, 5, False), ((128, 64, 32), 8, 1, True)]


@pytest.mark.parametrize("shape,P1,P2,c2c", CASES)
@pytest.mark.parametrize("prec", ["double", "float"])
def test_plan_algebra_matches_reference_formulas(shape, P1, P2, c2c, prec):
    """partition tables, block sizes/offsets, domain size and all-to-all byte tables for every
    rank == oracle restatement of mpicufft_pencil_opt1.cpp:67-93, 203-209, 265-320"""
    esz = 16 if prec == "double" else 8
    world = dfft.Comm.local(P1 * P2) if P1 * P2 > 1 else None
    opl = orc.PencilPlan(*shape, P1, P2, c2c)
    for r in range(P1 * P2):
        pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), world, precision=prec, rank=r)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(P1, P2), allocate=False, c2c=c2c)
        assert (pl.getInSize(), pl.getInStart()) == opl.in_block(r)
        assert (pl.getOutSize(), pl.getOutStart()) == opl.out_block(r)
        assert pl.getRank() == r and pl.getWorldSize() == P1 * P2
        dom = opl.domain_elems(r) * esz
        assert dom <= pl.getDomainSize() < dom + 256
        nexch = (P1 > 1) + (P2 > 1)
        assert pl.getWorkSizeDevice() == pl.getDomainSize() * (nexch + 1)
        for which in (1, 2):
            assert pl.getExchangeTables(which) == [[v * esz for v in t] for t in opl.exchange_tables(r, which)]


def test_init_errors():
    pl = dfft.MPIcuFFT_Slab(dfft.Configurations())
    with pytest.raises(dfft.DfftError, match="Invalid Input Partition"):
        pl.initFFT(dfft.GlobalSize(16, 16, 16), dfft.Slab_Partition(2), allocate=False)
    with pytest.raises(dfft.DfftError, match="unsupported"):
        pl.initFFT(dfft.GlobalSize(16, 16, 3000), dfft.Slab_Partition(1), allocate=False, c2c=True)
    world = dfft.Comm.local(4)
    ps = dfft.MPIcuFFT_Slab_Opt1(dfft.Configurations(), world, rank=0)
    with pytest.raises(dfft.DfftError, match="slab"):
        ps.initFFT(dfft.GlobalSize(16, 16, 16), dfft.Pencil_Partition(2, 2), allocate=False)
    with pytest.raises(dfft.DfftError, match="not initialised"):
        dfft.MPIcuFFT_Pencil(dfft.Configurations()).getInSize()
    with pytest.raises(dfft.DfftError):   # no GPU here: executing must fail loudly, never fall back
        pl2 = dfft.MPIcuFFT_Pencil(dfft.Configurations())
        pl2.initFFT(dfft.GlobalSize(16, 16, 16), dfft.Pencil_Partition(1, 1), allocate=False, c2c=True)
        pl2.execC2C(1, 1)


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under distributedfft_amd/ may reference it"""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "distributedfft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"\boracle\b|numpy\.fft|np\.fft|hipfft|rocfft", txt, re.I):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
