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
        for n in (4096, 8192):      # sub-tile workgroups: a few lines of a tile per workgroup
            info = dfft.kernel_info(n, prec)
            assert info and info["threads"] <= 512 and info["lds_bytes"] <= 80 * 1024
            assert info["threads"] * info["points_per_thread"] == n * info["lines_per_workgroup"]
        # lengths 2^a 3^b 5^c 7^d with a configuration in csrc/kernels_mixed.inc run the native chain too
        inc = open(os.path.join(os.path.dirname(dfft.__file__), "csrc", "kernels_mixed.inc")).read()
        tag = "F64" if prec == "double" else "F32"
        mixed = sorted({int(m) for m in re.findall(r"using %s_M(\d+) =" % tag, inc)})
        assert len(mixed) >= 40 and 1000 in mixed and 1536 in mixed and 2000 in mixed
        for n in mixed:
            info = dfft.kernel_info(n, prec)
            assert info and info["threads"] <= 1024 and info["lds_bytes"] <= 160 * 1024
            assert info["threads"] * info["points_per_thread"] == n * info["lines_per_workgroup"], (n, info)
        # every other length runs through Bluestein on the next power of two >= 2N-1 (N <= 4096)
        assert dfft.kernel_info(3, prec)["points_per_thread"] == 8 and dfft.kernel_info(1023, prec)["threads"] > 0
        assert dfft.kernel_info(1025, prec)["threads"] > 0 and dfft.kernel_info(4095, prec)["threads"] > 0
        # beyond that: two-level lines N = N1*N2 (each factor within reach of the generic kernel), and for what does not split -- a prime
        # above 4096, twice such a prime -- Bluestein over a two-level padded length; nothing beyond 2^24 points
        assert dfft.kernel_info(16384, prec)["threads"] > 0 and dfft.kernel_info(5000, prec)["threads"] > 0
        assert dfft.kernel_info(4099, prec)["threads"] > 0 and dfft.kernel_info(2 * 4099, prec)["threads"] > 0
        assert dfft.axis_plan_info(4099, prec) == {"kind": "long_bluestein", "M": 16384, "levels": [(128, 128, False), (128, 128, False)]}
        assert dfft.kernel_info((1 << 24) + 1, prec) is None


def test_axis_plans_cover_every_length_the_factors_allow():
    """dfft_axis_plan_info: native chain / Bluestein / two levels / long Bluestein.  Every length up to 4096 has a plan, powers of two
    up to 8192 have the native one, beyond that the lengths N1*N2 whose factors are each within reach of the generic kernel run in two
    levels and everything else up to 2^23 as Bluestein over a two-level padded length
    (the reference takes any length through cufftMakePlanMany64, mpicufft_pencil_opt1.cpp:165-197)."""
    def reach(f):      # one launch of the generic kernel: plain power of two <= 8192, or Bluestein with 2f-1 <= 8192
        return f >= 2 and ((f & (f - 1)) == 0 and f <= 8192 or 2 * f - 1 <= 8192)
    for prec in ("double", "float"):
        for n in list(range(2, 4097)) + [8192]:
            info = dfft.axis_plan_info(n, prec)
            assert info and info["kind"] in ("native", "bluestein"), n      # no regression: these keep their one-launch plans
        for n in list(range(4097, 4400)) + [5000, 6000, 8190, 8194, 9999, 10000, 12288, 16384, 65536, 100000, 131072, 3 * 4093,
                                            4099, 2 * 4099, 4093 * 4093, 4096 * 4096, 4096 * 4096 + 2, 1 << 25]:
            info = dfft.axis_plan_info(n, prec)
            splits = [d for d in range(2, int(n ** 0.5) + 1) if n % d == 0 and reach(d) and reach(n // d)] if n <= 1 << 24 else []
            # what does not split -- a prime above 4096 (4099), twice one (8198), 4093 x 4093 ... -- runs Bluestein over a two-level
            # padded length as long as that stays within 2^24 points
            pad = 1 << (2 * n - 2).bit_length()
            assert (info is not None) == (bool(splits) or pad <= 1 << 24), n
            if info and splits:
                assert info["kind"] == "two_level" and info["M"] == 0
                (n1, m1, b1), (n2, m2, b2) = info["levels"]
                assert n1 * n2 == n and reach(n1) and reach(n2)
                for f, m, b in info["levels"]:
                    assert (m == f and f & (f - 1) == 0) if not b else (m & (m - 1) == 0 and m >= 2 * f - 1 and m < 2 * (2 * f - 1))
            elif info:
                assert info["kind"] == "long_bluestein" and info["M"] == pad
                (n1, m1, b1), (n2, m2, b2) = info["levels"]
                assert n1 * n2 == pad and not b1 and not b2 and m1 == n1 and m2 == n2 and max(n1, n2) <= 8192
        # the option / variant that forces two levels: every composite length splits, primes keep their plan
        for n in (4, 6, 15, 64, 1000, 1024, 1155, 4096):
            info = dfft.axis_plan_info(n, prec, two_level=1)
            assert info["kind"] == "two_level" and info["levels"][0][0] * info["levels"][1][0] == n
        assert dfft.axis_plan_info(127, prec, two_level=1)["kind"] == "bluestein" and dfft.axis_plan_info(2, prec, two_level=1)["kind"] == "native"


def test_in_register_butterflies_on_the_host(tmp_path):
    """tests/cpp/butterfly_check.hip: the mixed-radix butterflies of fft_pass.hip.h (Dif, dft_prime, the compile-time
    trigonometry, the slot -> output index map) compiled for the host and checked against a long-double DFT"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(__file__), "cpp", "butterfly_check.hip")
    exe = str(tmp_path / "butterfly_check")
    subprocess.run([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", src, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


def test_whole_stockham_chain_emulated_on_the_host(tmp_path):
    """tests/cpp/chain_check.hip: the kernel's own pass_compute / lds_scatter / lds_gather and output index map, compiled
    for the host and driven for every thread of a workgroup (an array for LDS, loop boundaries for barriers), against a
    long-double DFT -- every generated mixed-radix configuration of both precisions plus a few powers of two"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(__file__), "cpp", "chain_check.hip")
    builds = []
    for tag, flag in (("f64", []), ("f32", ["-DCHAIN_F32"])):
        exe = str(tmp_path / ("chain_check_" + tag))
        builds.append((exe, subprocess.Popen([hipcc, "-O1", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", *flag, src, "-o", exe])))
    runs = []
    for exe, proc in builds:
        assert proc.wait() == 0
        runs.append(subprocess.Popen([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r in runs:
        out = r.communicate()[0]
        assert r.returncode == 0 and "ALL OK" in out, out[-2000:]


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


def _split(n, p):
    sizes = [n // p + (1 if i < n % p else 0) for i in range(p)]
    return sizes, [sum(sizes[:i]) for i in range(p)]


@pytest.mark.parametrize("shape,P1,P2,c2c", CASES[:8])
def test_partition_dimensions_match_reference_tables(shape, P1, P2, c2c):
    """getPartitionDimensions (include/mpicufft_pencil.hpp:112-116) == the three tables of
    src/pencil/mpicufft_pencil_opt1.cpp:70-93, restated here entry by entry"""
    Nx, Ny, Nz = shape
    Nzc = Nz if c2c else Nz // 2 + 1
    world = dfft.Comm.local(P1 * P2) if P1 * P2 > 1 else None
    for r in (0, P1 * P2 - 1):
        pl = dfft.MPIcuFFT_Pencil(dfft.Configurations(), world, rank=r)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(P1, P2), allocate=False, c2c=c2c)
        inp, tr, out = pl.getPartitionDimensions()
        assert (inp.size_x, inp.start_x) == _split(Nx, P1) and (inp.size_y, inp.start_y) == _split(Ny, P2)
        assert (inp.size_z, inp.start_z) == ([Nz], [0])
        assert (tr.size_x, tr.start_x) == _split(Nx, P1) and (tr.size_y, tr.start_y) == ([Ny], [0])
        assert (tr.size_z, tr.start_z) == _split(Nzc, P2)
        assert (out.size_x, out.start_x) == ([Nx], [0]) and (out.size_y, out.start_y) == _split(Ny, P1)
        assert (out.size_z, out.start_z) == _split(Nzc, P2)
        i, j = r // P2, r % P2
        assert pl.getInSize() == (inp.size_x[i], inp.size_y[j], inp.size_z[0])
        assert pl.getOutSize() == (out.size_x[0], out.size_y[i], out.size_z[j])
        assert pl.getOutStart() == (0, out.start_y[i], out.start_z[j])
        d = dfft.Partition_Dimensions()
        d.size_x, d.size_y, d.size_z = inp.size_x, inp.size_y, inp.size_z
        d.computeOffsets()
        assert (d.start_x, d.start_y, d.start_z) == (inp.start_x, inp.start_y, inp.start_z)


def test_class_hierarchy_mirrors_the_reference():
    """include/mpicufft_pencil_opt1.hpp:23, mpicufft_slab_opt1.hpp:70, mpicufft_slab_z_then_yx_opt1.hpp:22: the Opt1
    classes derive from their opt0 class (the reference's tests hold them through opt0 pointers)"""
    assert issubclass(dfft.MPIcuFFT_Pencil_Opt1, dfft.MPIcuFFT_Pencil)
    assert issubclass(dfft.MPIcuFFT_Slab_Opt1, dfft.MPIcuFFT_Slab)
    assert issubclass(dfft.MPIcuFFT_Slab_Z_Then_YX_Opt1, dfft.MPIcuFFT_Slab_Z_Then_YX)
    hdr = open(os.path.join(ROOT, "include", "mpicufft_amd.hpp")).read()
    for derived, base in (("MPIcuFFT_Pencil_Opt1", "MPIcuFFT_Pencil<T>"), ("MPIcuFFT_Slab_Opt1", "MPIcuFFT_Slab<T>"),
                          ("MPIcuFFT_Slab_Z_Then_YX_Opt1", "MPIcuFFT_Slab_Z_Then_YX<T>")):
        assert re.search(r"class %s : public %s" % (derived, re.escape(base)), hdr), derived


def test_options_and_reinit():
    """named options (dfft_set_option); the pipeline depth may change between two initFFT calls of one plan"""
    world = dfft.Comm.local(2)
    pl = dfft.MPIcuFFT_Slab_Opt1(dfft.Configurations(), world, rank=0)
    pl.setPipelineChunks(2)
    pl.initFFT(dfft.GlobalSize(32, 32, 32), dfft.Slab_Partition(2), allocate=False, c2c=True)
    assert pl.getPipelineChunks() == 2
    pl.setPipelineChunks(4)                      # (was refused after initFFT in round 1)
    assert pl.getPipelineChunks() == 2           # ... and applies at the next initFFT
    pl.initFFT(dfft.GlobalSize(32, 32, 32), dfft.Slab_Partition(2), allocate=False, c2c=True)
    assert pl.getPipelineChunks() == 4
    pl.setOption("pipeline_chunks", 3)
    assert pl.getOption("pipeline_chunks") == 3 and pl.getOption("mirror_inverse") == 0
    for key in ("mirror_inverse", "point_tables", "shift", "debug_skip", "variant_fx", "order_iz"):
        pl.setOption(key, 1)
        assert pl.getOption(key) == 1
    with pytest.raises(dfft.DfftError, match="unknown option"):
        pl.setOption("no_such_knob", 1)
    assert pl.getOption("no_such_knob") == -1


def test_r2c_with_two_point_z_axis_plans():
    """an R2C plan with Nz == 2 must plan its z axis for the Bluestein real modes (it used to pick the plain
    complex chain and fault at the first launch)"""
    pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations())
    pl.initFFT(dfft.GlobalSize(8, 8, 2), dfft.Pencil_Partition(1, 1), allocate=False)
    assert pl.getOutSize() == (8, 8, 2)


def test_init_errors():
    pl = dfft.MPIcuFFT_Slab(dfft.Configurations())
    with pytest.raises(dfft.DfftError, match="Invalid Input Partition"):
        pl.initFFT(dfft.GlobalSize(16, 16, 16), dfft.Slab_Partition(2), allocate=False)
    with pytest.raises(dfft.DfftError, match="unsupported"):
        pl.initFFT(dfft.GlobalSize(16, 16, (1 << 24) + 1), dfft.Slab_Partition(1), allocate=False, c2c=True)      # more than 2^24 points on a line
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


@pytest.mark.parametrize("shape,P1,P2,c2c", [((64, 32, 16), 4, 2, False), ((16, 16, 16), 3, 2, True), ((30, 20, 18), 2, 3, False),
                                            ((1024, 1024, 1024), 2, 4, True), ((128, 64, 32), 8, 1, True), ((9, 7, 10), 1, 3, True)])
@pytest.mark.parametrize("chunks", [1, 3, 4])
def test_pipeline_tables_are_consistent_across_ranks(shape, P1, P2, c2c, chunks):
    """the chunked all-to-all tables every rank uses: (1) what rank r sends to peer q in chunk c is
    exactly what q expects from r, (2) chunks add up to the reference's per-peer message
    (mpicufft_pencil_opt1.cpp:269-273, 315-319), (3) blocks tile the buffers without overlap"""
    esz = 16
    P = P1 * P2
    world = dfft.Comm.local(P) if P > 1 else None
    plans = []
    for r in range(P):
        pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), world, precision="double", rank=r)
        pl.setPipelineChunks(chunks)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(P1, P2), allocate=False, c2c=c2c)
        plans.append(pl)
    C = plans[0].getPipelineChunks()
    assert all(pl.getPipelineChunks() == C for pl in plans) and 1 <= C <= chunks
    for direction in (dfft.FORWARD, dfft.INVERSE):
        for which in (1, 2):
            n = P2 if which == 1 else P1
            for r, pl in enumerate(plans):
                i, j = r // P2, r % P2
                group = [i * P2 + q for q in range(P2)] if which == 1 else [q * P2 + j for q in range(P1)]
                me = j if which == 1 else i
                tot_s, tot_r, spans_s, spans_r = [0] * n, [0] * n, [], []
                for c in range(C):
                    sc, sd, rc, rd = pl.getPipelineTables(direction, which, c)
                    for q in range(n):
                        psc, psd, prc, prd = plans[group[q]].getPipelineTables(direction, which, c)
                        assert sc[q] == prc[me] and rc[q] == psc[me]
                        tot_s[q] += sc[q]
                        tot_r[q] += rc[q]
                        if sc[q]:
                            spans_s.append((sd[q], sd[q] + sc[q]))
                        if rc[q]:
                            spans_r.append((rd[q], rd[q] + rc[q]))
                ref = pl.getExchangeTables(which)
                want_s, want_r = (ref[0], ref[2]) if direction == dfft.FORWARD else (ref[2], ref[0])
                assert tot_s == want_s and tot_r == want_r
                for spans in (spans_s, spans_r):
                    spans.sort()
                    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "blocks overlap"
                    assert not spans or spans[-1][1] <= pl.getDomainSize()
                    assert sum(b - a for a, b in spans) % esz == 0


@pytest.mark.parametrize("shape,P,c2c", [((16, 8, 16), 2, True), ((30, 20, 18), 3, False), ((64, 6, 32), 8, True),
                                         ((1024, 1024, 1024), 8, False)])
@pytest.mark.parametrize("chunks", [1, 3, 4])
def test_z_then_yx_plan_algebra_and_tables(shape, P, c2c, chunks):
    """slab sequence Z_Then_YX: sizes/offsets (mpicufft_slab_z_then_yx.cpp:85-106, hpp:41-44), the
    per-peer byte counts (:190-196) and the chunked tables built from them"""
    Nx, Ny, Nz = shape
    Nzc = Nz if c2c else Nz // 2 + 1
    esz = 16
    world = dfft.Comm.local(P)
    plans = []
    for r in range(P):
        pl = dfft.MPIcuFFT_Slab_Z_Then_YX(dfft.Configurations(), world, precision="double", rank=r)
        pl.setPipelineChunks(chunks)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Slab_Partition(P), allocate=False, c2c=c2c)
        plans.append(pl)

    def split(n):
        return [n // P + (1 if q < n % P else 0) for q in range(P)]

    xs, zs = split(Nx), split(Nzc)
    x0 = [sum(xs[:q]) for q in range(P)]
    z0 = [sum(zs[:q]) for q in range(P)]
    C = plans[0].getPipelineChunks()
    for r, pl in enumerate(plans):
        assert list(pl.getInSize()) == [xs[r], Ny, Nz] and list(pl.getInStart()) == [x0[r], 0, 0]
        assert list(pl.getOutSize()) == [Nx, Ny, zs[r]] and list(pl.getOutStart()) == [0, 0, z0[r]]
        assert pl.getDomainSize() >= esz * max(xs[r] * Ny * Nzc, Nx * Ny * zs[r])
        assert pl.getWorkSizeDevice() == 2 * pl.getDomainSize()
        sc, sd, rc, rd = pl.getExchangeTables(2)
        assert sc == [esz * zs[q] * Ny * xs[r] for q in range(P)]
        assert sd == [esz * z0[q] * Ny * xs[r] for q in range(P)]
        assert rc == [esz * zs[r] * Ny * xs[q] for q in range(P)]
        assert rd == [esz * zs[r] * Ny * x0[q] for q in range(P)]
        for direction in (dfft.FORWARD, dfft.INVERSE):
            tot_s, tot_r, spans_s, spans_r = [0] * P, [0] * P, [], []
            for c in range(C):
                tsc, tsd, trc, trd = pl.getPipelineTables(direction, 2, c)
                for q in range(P):
                    psc, _, prc, _ = plans[q].getPipelineTables(direction, 2, c)
                    assert tsc[q] == prc[r] and trc[q] == psc[r]
                    tot_s[q] += tsc[q]
                    tot_r[q] += trc[q]
                    if tsc[q]:
                        spans_s.append((tsd[q], tsd[q] + tsc[q]))
                    if trc[q]:
                        spans_r.append((trd[q], trd[q] + trc[q]))
            assert (tot_s, tot_r) == ((sc, rc) if direction == dfft.FORWARD else (rc, sc))
            for spans in (spans_s, spans_r):
                spans.sort()
                assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "blocks overlap"
                assert spans[-1][1] <= pl.getDomainSize()
    # a single rank is the local 3-D transform with the usual output block
    one = dfft.MPIcuFFT_Slab_Z_Then_YX(dfft.Configurations(), precision="double")
    one.initFFT(dfft.GlobalSize(*shape), dfft.Slab_Partition(1), allocate=False, c2c=c2c)
    assert list(one.getOutSize()) == [Nx, Ny, Nzc]


@pytest.mark.parametrize("shape,P,c2c", [((16, 8, 16), 2, False), ((30, 20, 18), 3, False), ((64, 16, 32), 8, True),
                                         ((1024, 1024, 1024), 8, False)])
def test_y_then_zx_plan_algebra_and_tables(shape, P, c2c):
    """slab sequence Y_Then_ZX: Hermitian axis y, output [Nx][(Ny/2+1)/P][Nz]
    (mpicufft_slab_y_then_zx.cpp:84-108, hpp:40-43), exchange counts (:309-319)"""
    Nx, Ny, Nz = shape
    Nyc = Ny if c2c else Ny // 2 + 1
    esz = 16
    world = dfft.Comm.local(P)
    plans = []
    for r in range(P):
        pl = dfft.MPIcuFFT_Slab_Y_Then_ZX(dfft.Configurations(), world, precision="double", rank=r)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Slab_Partition(P), allocate=False, c2c=c2c)
        plans.append(pl)

    def split(n):
        return [n // P + (1 if q < n % P else 0) for q in range(P)]

    xs, yo = split(Nx), split(Nyc)
    C = plans[0].getPipelineChunks()
    for r, pl in enumerate(plans):
        assert list(pl.getInSize()) == [xs[r], Ny, Nz] and list(pl.getInStart()) == [sum(xs[:r]), 0, 0]
        assert list(pl.getOutSize()) == [Nx, yo[r], Nz] and list(pl.getOutStart()) == [0, sum(yo[:r]), 0]
        assert pl.getDomainSize() >= esz * max(xs[r] * Nyc * Nz, Nx * yo[r] * Nz)
        sc, sd, rc, rd = pl.getExchangeTables(2)
        assert sc == [esz * Nz * yo[q] * xs[r] for q in range(P)] and rd == [esz * Nz * yo[r] * sum(xs[:q]) for q in range(P)]
        tot_s, tot_r = [0] * P, [0] * P
        for c in range(C):
            tsc, tsd, trc, trd = pl.getPipelineTables(dfft.FORWARD, 2, c)
            for q in range(P):
                psc, _, prc, _ = plans[q].getPipelineTables(dfft.FORWARD, 2, c)
                assert tsc[q] == prc[r] and trc[q] == psc[r]
                tot_s[q] += tsc[q]
                tot_r[q] += trc[q]
        assert tot_s == sc and tot_r == rc
        with pytest.raises(dfft.DfftError):
            pl.getPipelineTables(dfft.INVERSE, 2, 0)


def test_transport_probe_failure_paths_without_a_gpu():
    """the child-process probe of the native RCCL transport: a hang is cut off by the timeout, an
    error is reported as failure (here: no GPU), and only a clean exit counts as success"""
    import sys as _sys

    from distributedfft_amd import torch_transport as tt
    assert tt._probe_native(1, 1, timeout=1, cmd=[_sys.executable, "-c", "import time; time.sleep(30)"]) is False
    assert tt._probe_native(1, 1, timeout=60, cmd=[_sys.executable, "-c", "raise SystemExit(0)"]) is True
    import torch
    if not torch.cuda.is_available():
        assert tt._probe_native(1, 1, timeout=120) is False


def test_bench_dry_run_plans_c4_and_c5_for_all_ranks():
    """bench.py --dry-run builds BASELINE's multi-GPU plans for all 8 ranks on a host without a GPU: C4 = 2 GiB
    domain, C5 = 8 GiB domain with a 3-slice work area, both far below the 288 GB of one MI355X"""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])["dry_run"]
    assert d["C4"]["domain_GiB"] == 2.0 and d["C4"]["work_GiB"] == 6.0 and d["C4"]["fits_288_GB"]
    assert d["C5"]["domain_GiB"] == 8.0 and d["C5"]["work_GiB"] == 24.0 and d["C5"]["fits_288_GB"]
    assert d["C5"]["per_gpu_total_GiB (in + out + back + work)"] == 48.0
    assert d["C3"]["partition"] == "2x1" and d["C4-slab"]["work_GiB"] == 4.0


def test_bench_partitions_and_xgmi_model():
    """bench.py helpers (no GPU): the headline decomposition at N > 1 is the one BASELINE.json names, and the xGMI model
    charges every link of an exchange group with 1/P of the local volume at 153 GB/s"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.choose_partition(8, "auto") == (2, 4) and b.choose_partition(4, "auto") == (2, 2) and b.choose_partition(2, "auto") == (2, 1)
    assert b.choose_partition(8, "slab") == (8, 1) and b.choose_partition(8, "pencil") == (2, 4)
    m = b.xgmi_model(16, 1024, 8, 2, 4)
    vol = 16 * 1024 ** 3 / 8
    assert m["exchange 1"]["links"] == 3 and m["exchange 1"]["bytes_per_link"] == vol / 4
    assert m["exchange 2"]["links"] == 1 and m["exchange 2"]["bytes_per_link"] == vol / 2
    assert abs(m["exchange 2"]["predicted_ms"] - vol / 2 / 153e9 * 1e3) < 1e-3
    s8 = b.xgmi_model(16, 1024, 8, 8, 1)
    assert set(s8) == {"exchange 2"} and s8["exchange 2"]["links"] == 7


def test_comm_options_and_variant_range():
    """dfft_comm_set_option: transports without the option say so; variant_* outside 0..15 is rejected (the dispatch key has
    four bits for the variant)"""
    world = dfft.Comm.local(2)
    with pytest.raises(dfft.DfftError):
        world.setOption("dup_channel", 1)
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    with pytest.raises(dfft.DfftError, match="variant"):
        plan.setOption("variant_fy", 16)
    plan.setOption("variant_fy", 15)
    plan.setOption("uniform_tables", 0)
    assert plan.getOption("uniform_tables") == 0


def test_bench_xgmi_model_with_the_relay():
    """bench.py's model of the exchanges (bytes per link / 153 GB/s) and of their two-hop relay: C4 = 1024^3 fp64 on pencil 2 x 4 --
    exchange 2 moves 1 GiB over ONE link (7.0 ms) or, relayed, 2 x 1/8 GiB over each of seven (1.75 ms); exchange 1 3.5 -> 2.6 ms;
    a slab exchange spans the world: nothing to relay through"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    m = bench.xgmi_model(16, 1024, 8, 2, 4)
    assert m["exchange 2"]["links"] == 1 and abs(m["exchange 2"]["predicted_ms"] - 7.018) < 0.01
    assert m["exchange 2"]["relay"] == {"links": 7, "phases": 2, "pieces_per_link_per_phase": 1, "bytes_per_link_per_phase": 2 ** 27, "predicted_ms": 1.754}
    assert m["exchange 1"]["links"] == 3 and abs(m["exchange 1"]["predicted_ms"] - 3.509) < 0.01
    assert m["exchange 1"]["relay"]["phases"] == 2 and m["exchange 1"]["relay"]["pieces_per_link_per_phase"] == 3
    assert abs(m["exchange 1"]["relay"]["predicted_ms"] - 2.632) < 0.01
    s = bench.xgmi_model(16, 1024, 8, 8, 1)
    assert list(s) == ["exchange 2"] and "relay" not in s["exchange 2"] and s["exchange 2"]["links"] == 7
    assert bench.choose_partition(8, "auto") == (2, 4) and bench.choose_partition(4, "auto") == (2, 2) and bench.choose_partition(2, "auto") == (2, 1)


def test_pass_choices_follow_the_role_rules():
    """dfft_get_pass_choices on plans built without a GPU: the kernel configuration of every pass by the rules of dfft_init (the role
    numbers of csrc/cfg_f64.hip.h / cfg_f32.hip.h) -- what dfft_tune_variants starts from and reports after it ran"""
    import distributedfft_amd as dfft

    def choices(prec, n, P1, P2, c2c=True, options=None, rank=0):
        world = dfft.Comm.local(P1 * P2) if P1 * P2 > 1 else None
        pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), world, precision=prec, rank=rank)
        for k, v in (options or {}).items():
            pl.setOption(k, v)
        pl.initFFT(dfft.GlobalSize(n, n, n), dfft.Partition(P1, P2), allocate=False, c2c=c2c)
        return {k: v[0] for k, v in pl.getPassChoices().items()}

    # fp64 1024^3 on 2 x 4: streaming z passes, the strided-read configuration on the inverse x pass, streaming inverse y
    assert choices("double", 1024, 2, 4) == {"fz": 3, "fy": 0, "fx": 0, "ix": 1, "iy": 3, "iz": 3}
    # ... R2C: 129-wide rows are not whole tiles -- the inverse x pass keeps the default (8 lines, two workgroups per CU)
    assert choices("double", 1024, 2, 4, c2c=False)["ix"] == 0
    # ... x-contiguous spectrum: both x passes have natural lines on one side
    assert choices("double", 1024, 2, 4, options={"spectral_layout": 1}) == {"fz": 3, "fy": 0, "fx": 3, "ix": 3, "iy": 3, "iz": 3}
    # fp32 2048^3 on 2 x 4: natural-line z passes (4 / 5), the streaming tiled configuration on forward y and inverse x (round 5:
    # what the tuner kept on every 2048^3 plan), whole-line transposed stores on the inverse y pass (7)
    assert choices("float", 2048, 2, 4) == {"fz": 4, "fy": 9, "fx": 6, "ix": 9, "iy": 7, "iz": 5}
    assert choices("float", 2048, 2, 4, options={"spectral_layout": 1}) == {"fz": 4, "fy": 9, "fx": 5, "ix": 1, "iy": 7, "iz": 5}
    # one rank keeps the tiled configuration on its y pass (the rule is for multi-rank plans)
    assert choices("float", 2048, 1, 1, options={"single_order": 0})["fy"] == 6
    # a pinned pass stays pinned
    assert choices("double", 1024, 2, 4, options={"variant_ix": 2})["ix"] == 2


# ------------------------------------------------------------------------------------------
# the two libraries: libdfft_amd.so (powers of two up to 8192: every BASELINE configuration) and libdfft_amd_any.so (every other
# length), which the core opens at the first plan that needs it (csrc/any_loader.hip, csrc/any_exports.hip)
# ------------------------------------------------------------------------------------------
ANY_SYMBOLS = ["dfft_any_launch_mixed_f64", "dfft_any_launch_mixed_f32", "dfft_any_mixed_info_f64", "dfft_any_mixed_info_f32",
               "dfft_any_launch_rmixed_f64", "dfft_any_launch_rmixed_f32", "dfft_any_rmixed_info_f64", "dfft_any_rmixed_info_f32",
               "dfft_any_launch_bluestein_f64", "dfft_any_launch_bluestein_f32"]


def test_second_library_exports_what_the_core_looks_up():
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "distributedfft_amd", "libdfft_amd_any.so")
    assert os.path.exists(path), "libdfft_amd_any.so is not built: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    for name in ANY_SYMBOLS:
        assert hasattr(lib, name), name
    # every name the loader asks for is one of these (the two files cannot drift apart unnoticed)
    src = open(os.path.join(root, "distributedfft_amd", "csrc", "any_loader.hip")).read()
    import re
    assert sorted(set(re.findall(r'sym\("(dfft_any_\w+)"\)', src))) == sorted(ANY_SYMBOLS)


def test_plans_outside_the_core_fail_loudly_without_the_second_library():
    """DFFT_ANY_LIBRARY names a file that does not exist: powers of two plan as always, any other length refuses to initialise and
    says which library is missing -- no fallback, no crash at the first launch"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import distributedfft_amd as dfft\n"
        "pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations())\n"
        "pl.initFFT(dfft.GlobalSize(64, 32, 16), dfft.Pencil_Partition(1, 1), False)\n"
        "print('pow2 ok', pl.getDomainSize())\n"
        "for shape in ((12, 16, 16), (16, 17, 16), (16, 16, 10000), (16384, 4, 4)):\n"
        "    try:\n"
        "        pl.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), False, c2c=True)\n"
        "        print('UNEXPECTED', shape)\n"
        "    except dfft.DfftError as e:\n"
        "        print('refused', shape, 'libdfft_amd_any.so' in str(e) and 'cannot load' in str(e))\n")
    env = dict(os.environ, DFFT_ANY_LIBRARY="/nonexistent/libdfft_amd_any.so", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().split("\n")
    assert lines[0].startswith("pow2 ok")
    assert [ln for ln in lines[1:]] == [f"refused {s} True" for s in ((12, 16, 16), (16, 17, 16), (16, 16, 10000), (16384, 4, 4))], out.stdout
    # and with the library in place the same lengths plan
    env.pop("DFFT_ANY_LIBRARY")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.count("UNEXPECTED") == 4, out.stdout + out.stderr
