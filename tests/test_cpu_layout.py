"""Every pass descriptor, segment table, per-point table and chunked exchange table of a plan,
executed on the CPU by tests/layout_sim.py (numpy transforms + the documented address forms) and
compared with the transform of the global array -- the GPU-free check of the plan wiring
(build_pipeline / build_pipeline_zyx / build_pipeline_yzx in distributedfft_amd/csrc/dfft.hip).
The kernels themselves are covered by the -m gpu parity tests."""
import numpy as np
import pytest

import distributedfft_amd as dfft
from layout_sim import World

RNG = np.random.default_rng(7)


def global_field(shape, c2c):
    g = RNG.uniform(0, 255, shape)
    return g + 1j * RNG.uniform(0, 255, shape) if c2c else g


def local_inputs(world, g):
    ins = []
    for pl in world.plans:
        s, o = pl.getInSize(), pl.getInStart()
        blk = g[o[0]:o[0] + s[0], o[1]:o[1] + s[1], :]
        ins.append(np.ascontiguousarray(blk).ravel().astype(np.complex128 if world.c2c else np.float64))
    return ins


def check_spectrum(world, outs, want):
    scale = np.max(np.abs(want))
    for pl, out in zip(world.plans, outs):
        s, o = pl.getOutSize(), pl.getOutStart()
        got = out[:s[0] * s[1] * s[2]].reshape(s)
        ref = want[o[0]:o[0] + s[0], o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
        assert np.max(np.abs(got - ref)) / scale < 1e-12


def check_round_trip(world, backs, ins):
    n3 = float(np.prod(world.shape))
    for b, x in zip(backs, ins):
        assert np.max(np.abs(b / n3 - x)) / 255.0 < 1e-12


DEFAULT = [((12, 10, 14), 2, 2), ((9, 7, 10), 3, 2), ((16, 8, 8), 2, 1), ((8, 8, 16), 1, 2), ((6, 5, 9), 1, 1), ((10, 20, 18), 2, 3)]


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P1,P2", DEFAULT)
def test_default_sequence_descriptors(shape, P1, P2, c2c, chunks):
    w = World(dfft.MPIcuFFT_Pencil_Opt1, shape, P1, P2, c2c, chunks)
    g = global_field(shape, c2c)
    ins = local_inputs(w, g)
    outs = w.forward(ins)
    check_spectrum(w, outs, np.fft.fftn(g) if c2c else np.fft.rfftn(g))
    check_round_trip(w, w.inverse(outs), ins)


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P1,P2", DEFAULT)
def test_x_contiguous_spectrum_descriptors(shape, P1, P2, c2c, chunks):
    """option spectral_layout = 1: the forward x pass stores natural lines -- the spectrum block is [yo][zs][Nx], entry (kx, ky, kz) at
    the strides dfft_get_out_strides reports -- and the inverse x pass loads them; sizes, starts and exchange tables are the
    reference's.  One rank: the inverse runs the mirrored pass order."""
    w = World(dfft.MPIcuFFT_Pencil_Opt1, shape, P1, P2, c2c, chunks, options={"spectral_layout": 1})
    ref = World(dfft.MPIcuFFT_Pencil_Opt1, shape, P1, P2, c2c, chunks)
    g = global_field(shape, c2c)
    ins = local_inputs(w, g)
    outs = w.forward(ins)
    want = np.fft.fftn(g) if c2c else np.fft.rfftn(g)
    scale = np.max(np.abs(want))
    for pl, rp, out in zip(w.plans, ref.plans, outs):
        s, o, st = pl.getOutSize(), pl.getOutStart(), pl.getOutStrides()
        assert (s, o) == (rp.getOutSize(), rp.getOutStart()) and pl.getDomainSize() == rp.getDomainSize()
        assert pl.getExchangeTables(1) == rp.getExchangeTables(1) and pl.getExchangeTables(2) == rp.getExchangeTables(2)
        assert st == (1, s[2] * s[0], s[0]) and rp.getOutStrides() == (s[1] * s[2], s[2], 1)
        got = np.lib.stride_tricks.as_strided(out, shape=s, strides=tuple(16 * v for v in st))
        assert np.max(np.abs(got - want[o[0]:o[0] + s[0], o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < 1e-12
    check_round_trip(w, w.inverse(outs), ins)


def test_x_contiguous_spectrum_is_for_the_default_sequences():
    pl = dfft.MPIcuFFT_Slab_Z_Then_YX(dfft.Configurations(), dfft.Comm.local(2), precision="double", rank=0)
    pl.setOption("spectral_layout", 1)
    with pytest.raises(dfft.DfftError, match="spectral_layout"):
        pl.initFFT(dfft.GlobalSize(8, 8, 8), dfft.Partition(2, 1), allocate=False, c2c=True)
    pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), None, precision="double", rank=0)
    pl.setOption("spectral_layout", 2)
    with pytest.raises(dfft.DfftError, match="spectral_layout"):
        pl.initFFT(dfft.GlobalSize(8, 8, 8), dfft.Partition(1, 1), allocate=False, c2c=True)


@pytest.mark.parametrize("chunks", [1, 2])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P", [((12, 10, 14), 2), ((9, 6, 20), 3), ((16, 5, 16), 4)])
def test_z_then_yx_descriptors(shape, P, c2c, chunks):
    w = World(dfft.MPIcuFFT_Slab_Z_Then_YX, shape, P, 1, c2c, chunks)
    g = global_field(shape, c2c)
    ins = local_inputs(w, g)
    outs = w.forward(ins, "zyx")
    check_spectrum(w, outs, np.fft.fftn(g) if c2c else np.fft.rfftn(g))
    check_round_trip(w, w.inverse(outs, "zyx"), ins)


@pytest.mark.parametrize("chunks", [1, 2])
@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P", [((12, 10, 14), 2), ((9, 12, 7), 3), ((8, 6, 4), 1)])
def test_y_then_zx_descriptors(shape, P, c2c, chunks):
    w = World(dfft.MPIcuFFT_Slab_Y_Then_ZX, shape, P, 1, c2c, chunks)
    g = global_field(shape, c2c)
    outs = w.forward(local_inputs(w, g), "yzx")
    want = np.fft.fftn(g)
    check_spectrum(w, outs, want if c2c else want[:, :shape[1] // 2 + 1, :])


@pytest.mark.parametrize("c2c", [True, False])
@pytest.mark.parametrize("shape,P1,P2", [((12, 10, 14), 2, 2), ((9, 7, 10), 3, 2), ((6, 5, 9), 1, 1)])
@pytest.mark.parametrize("d", [1, 2])
def test_partial_transform_descriptors(shape, P1, P2, d, c2c):
    """execR2C/execC2R(out, in, d): stage layouts [xs][ys][Nzc] (d = 1) and [xs][Ny][zs] (d = 2)"""
    w = World(dfft.MPIcuFFT_Pencil, shape, P1, P2, c2c, 2)
    g = global_field(shape, c2c)
    ins = local_inputs(w, g)
    outs = w.partial(ins, d, dfft.FORWARD)
    gz = np.fft.fft(g, axis=2) if c2c else np.fft.rfft(g, axis=2)
    stage = gz if d == 1 else np.fft.fft(gz, axis=1)
    scale = np.max(np.abs(stage))
    spec = []
    for pl, out in zip(w.plans, outs):
        s, o, os_, oo = pl.getInSize(), pl.getInStart(), pl.getOutSize(), pl.getOutStart()
        if d == 1:
            ref = stage[o[0]:o[0] + s[0], o[1]:o[1] + s[1], :]
        else:
            ref = stage[o[0]:o[0] + s[0], :, oo[2]:oo[2] + os_[2]]
        got = out[:ref.size].reshape(ref.shape)
        assert np.max(np.abs(got - ref)) / scale < 1e-12
        spec.append(out)
    backs = w.partial(spec, d, dfft.INVERSE)
    norm = float(shape[2]) if d == 1 else float(shape[2] * shape[1])
    for b, x in zip(backs, ins):
        assert np.max(np.abs(b / norm - x)) / 255.0 < 1e-12


@pytest.mark.parametrize("cls,kind,shape,P1,P2,chunks,prec", [
    (dfft.MPIcuFFT_Pencil_Opt1, "default", (16, 16, 12), 4, 4, 2, "double"),      # 16 ranks
    (dfft.MPIcuFFT_Slab_Opt1, "default", (32, 32, 6), 8, 1, 4, "double"),          # 8 peers x 4 chunks = 32 segments
    (dfft.MPIcuFFT_Pencil_Opt1, "default", (12, 20, 36), 2, 2, 2, "float"),       # fp32: 16-line tiles
    (dfft.MPIcuFFT_Pencil_Opt1, "default", (9, 7, 10), 3, 2, 3, "float"),
    (dfft.MPIcuFFT_Slab_Z_Then_YX, "zyx", (32, 6, 40), 8, 1, 4, "double"),
    (dfft.MPIcuFFT_Slab_Z_Then_YX, "zyx", (12, 18, 34), 3, 1, 2, "float"),
])
@pytest.mark.parametrize("c2c", [True, False])
def test_many_ranks_max_segments_and_fp32_tiles(cls, kind, shape, P1, P2, chunks, prec, c2c):
    w = World(cls, shape, P1, P2, c2c, chunks, precision=prec)
    if (P1, chunks) == (8, 4):
        assert w.C == 4       # 32 segments on the gathered axis: the MAXSEG limit of the kernels
    g = global_field(shape, c2c)
    ins = local_inputs(w, g)
    outs = w.forward(ins, kind)
    check_spectrum(w, outs, np.fft.fftn(g) if c2c else np.fft.rfftn(g))
    check_round_trip(w, w.inverse(outs, kind), ins)


@pytest.mark.parametrize("prec", ["double", "float"])
@pytest.mark.parametrize("options", [{"single_order": 1}, {"single_order": 1, "single_layout": 0}, {"single_order": 1, "single_pad": 0},
                                     {"single_order": 1, "single_layout": 0, "single_pad": 384}, {"single_order": 0}, {}])
@pytest.mark.parametrize("shape", [(16, 8, 32), (6, 5, 9), (24, 16, 20), (9, 16, 33)])
def test_single_rank_complex_orders_and_padded_layouts(shape, options, prec):
    """one rank, complex: the z, x, y order (strided natural-line rows -> L1 -> padded L2 -> natural) in both L2 layouts
    and with different row paddings, and the z, y, x order it replaces; forward == fftn, inverse round trip"""
    w = World(dfft.MPIcuFFT_Pencil_Opt1, shape, 1, 1, True, precision=prec, options=options)
    assert w.single == (options.get("single_order", 0) == 1)
    if w.single:
        d = w.plans[0].debugPass("sx")
        esz = 16 if prec == "double" else 8
        pad = options.get("single_pad", 128) // esz
        TL = w.plans[0].getTileLines()
        row = (TL * shape[1] if options.get("single_layout", 1) == 1 else shape[2] * shape[1]) + pad
        assert d.SK == row and w.plans[0].debugPass("sy").IA == row
        assert w.plans[0].getWorkSizeDevice() >= w.plans[0].getDomainSize()
    g = global_field(shape, True)
    ins = local_inputs(w, g)
    outs = w.forward(ins)
    check_spectrum(w, outs, np.fft.fftn(g))
    check_round_trip(w, w.inverse(outs), ins)
