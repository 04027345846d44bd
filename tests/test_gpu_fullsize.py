"""BASELINE.json's configurations at FULL size on one MI355X (ranks > 1 run as virtual ranks that
share the GPU, like MPI ranks with cudaSetDevice(rank % dev_count),
tests/src/pencil/random_dist_3D.cu:175-177).

C2 256^3 and C3 512^3 are still small enough to compare every output point with the CPU oracle.
C4 1024^3 (pencil 2x4) is checked through size-independent properties: round trip (reference
testcase 3), spectrum entries against a direct DFT evaluated with torch in fp64, the DC term,
Parseval and linearity.  C5's 2048^3 fp32 grid needs 8 GPUs' worth of HBM; its axis length and
precision are exercised on a 2048 x 256 x 256 and a 1024^3 fp32 grid instead.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402

from parity_metric import CENTER, check_forward, entry_rel, forward_bound, record  # noqa: E402

CDT = {"double": torch.complex128, "float": torch.complex64}
RDT = {"double": torch.float64, "float": torch.float32}


def host_rms(a):
    """sqrt(mean |a|^2) of a (large) host array without temporaries"""
    v = a.reshape(-1)
    return math.sqrt(float(np.vdot(v, v).real) / v.size)


def make_world(shape, P1, P2, prec, c2c=True, seed=1234, center=False):
    """plans + device buffers for P1*P2 virtual ranks; input generated on the device: uniform[0, 255) like the reference's
    (tests/src/pencil/base.cu:45-53), or the same centred on zero (center=True: parity_metric.py)"""
    P = P1 * P2
    world = dfft.Comm.local(P) if P > 1 else None
    esz = 16 if prec == "double" else 8
    ranks = []
    for r in range(P):
        pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), world, precision=prec, rank=r)
        pl.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(P1, P2), True, c2c=c2c)
        size = pl.getInSize()
        n = size[0] * size[1] * size[2]
        g = torch.Generator(device="cuda")
        g.manual_seed(seed + r)
        if c2c:
            x = torch.view_as_complex(torch.rand((n, 2), dtype=RDT[prec], device="cuda", generator=g) * 255 - (CENTER if center else 0.0))
        else:
            x = torch.rand(n, dtype=RDT[prec], device="cuda", generator=g) * 255 - (CENTER if center else 0.0)
        out = torch.zeros(pl.getDomainSize() // esz, dtype=CDT[prec], device="cuda")
        ranks.append(dict(plan=pl, x=x.reshape(size), out=out, back=torch.zeros_like(x).reshape(size)))
    torch.cuda.synchronize()
    return ranks


def run_all(ranks, fn):
    with ThreadPoolExecutor(len(ranks)) as ex:
        list(ex.map(fn, ranks))
    torch.cuda.synchronize()


def spectrum_block(rk):
    s = rk["plan"].getOutSize()
    return rk["out"][:s[0] * s[1] * s[2]].reshape(s)


def direct_dft_entry(ranks, shape, k):
    """X[k] = sum_r x[r] exp(-2 pi i k.r/N) evaluated block by block in fp64 on the device"""
    tot = torch.zeros((), dtype=torch.complex128, device="cuda")
    for rk in ranks:
        size, start = rk["plan"].getInSize(), rk["plan"].getInStart()
        w = []
        for a in range(3):
            idx = torch.arange(start[a], start[a] + size[a], device="cuda", dtype=torch.float64)
            ang = -2.0 * math.pi * ((idx * k[a]) % shape[a]) / shape[a]
            w.append(torch.complex(torch.cos(ang), torch.sin(ang)))
        x = rk["x"].to(torch.complex128) if rk["x"].dtype != torch.complex128 else rk["x"]
        tot += torch.einsum("xyz,z,y,x->", x, w[2], w[1], w[0])
    return complex(tot.item())


def owner_entry(ranks, k):
    for rk in ranks:
        s, o = rk["plan"].getOutSize(), rk["plan"].getOutStart()
        if o[1] <= k[1] < o[1] + s[1] and o[2] <= k[2] < o[2] + s[2]:
            return complex(spectrum_block(rk)[k[0], k[1] - o[1], k[2] - o[2]].item())
    raise AssertionError("no owner")


@pytest.mark.parametrize("center", [False, True])
def test_c2_256_single_gpu_every_point_vs_oracle(center):
    shape = (256, 256, 256)
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=2) - (CENTER * (1 + 1j) if center else 0)
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), True, c2c=True)
    d_in = torch.from_numpy(g).cuda()
    d_out = torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    plan.execC2C(d_out, d_in, dfft.FORWARD)
    want = orc.fft3d_c2c(g, -1)
    got = d_out[:g.size].cpu().numpy().reshape(shape)
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 1e-11
    check_forward(got, want, "double", g.size, zero_mean=center, label=f"C2 256^3 fp64 one rank, every point, centred={center}")
    back = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    plan.execC2C(back, d_out, dfft.INVERSE)
    assert np.max(np.abs(back.cpu().numpy() / g.size - g)) / 255 < 1e-10


@pytest.mark.parametrize("center", [False, True])
def test_c3_512_slab_two_ranks_every_point_vs_oracle(center):
    shape = (512, 512, 512)
    ranks = make_world(shape, 2, 1, "double", center=center)
    g = np.empty(shape, dtype=np.complex128)
    for rk in ranks:
        s, o = rk["plan"].getInSize(), rk["plan"].getInStart()
        g[o[0]:o[0] + s[0], o[1]:o[1] + s[1], :] = rk["x"].cpu().numpy()
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["out"], rk["x"], dfft.FORWARD))
    want = orc.fft3d_c2c(g, -1)
    scale = np.max(np.abs(want))
    want_rms = host_rms(want)
    for r, rk in enumerate(ranks):
        s, o = rk["plan"].getOutSize(), rk["plan"].getOutStart()
        got = spectrum_block(rk).cpu().numpy()
        assert np.max(np.abs(got - want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]])) / scale < 1e-11
        check_forward(got, want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]], "double", want.size, want_rms=want_rms, zero_mean=center,
                      label=f"C3 512^3 fp64 slab 2, rank {r}, every point, centred={center}")
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["back"], rk["out"], dfft.INVERSE))
    for rk in ranks:
        assert float((rk["back"] / float(np.prod(shape)) - rk["x"]).abs().max()) / 255 < 1e-10


@pytest.mark.parametrize("P1,P2", [(2, 4), (8, 1), (1, 1)])
def test_c4_1024_fp64_properties(P1, P2):
    """1024^3 fp64 complex, pencil 2x4 (BASELINE C4), slab 8 and single rank"""
    shape = (1024, 1024, 1024)
    n3 = float(np.prod(shape))
    ranks = make_world(shape, P1, P2, "double")
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["out"], rk["x"], dfft.FORWARD))
    # known answers: direct DFT of a few entries (incl. DC, Nyquist corners and generic points); per entry: against the rms of
    # the spectrum, sqrt(sum |x|^2) by Parseval (the direct sum in fp64 is itself good to ~1e-15 * sqrt(N^3) relative to that)
    scale = abs(owner_entry(ranks, (0, 0, 0)))
    ex = sum(float((rk["x"].abs() ** 2).sum()) for rk in ranks)
    spec_rms = math.sqrt(ex)
    for k in [(0, 0, 0), (512, 512, 512), (1, 2, 3), (1023, 1, 640), (300, 777, 129), (17, 1000, 1023)]:
        want = direct_dft_entry(ranks, shape, k)
        got = owner_entry(ranks, k)
        assert abs(got - want) / scale < 1e-11, (k, got, want)
        assert entry_rel(got, want, spec_rms) < forward_bound("double", n3), (k, got, want)
        record(f"C4 1024^3 fp64 {P1}x{P2} entry {k} vs direct DFT", "double", int(n3), entry_rel(got, want, spec_rms), forward_bound("double", n3), abs(got - want) / scale)
    # Parseval: sum |X|^2 = N^3 sum |x|^2
    eX = sum(float((spectrum_block(rk).abs() ** 2).sum()) for rk in ranks)
    assert abs(eX / (n3 * ex) - 1.0) < 1e-12
    # round trip (reference testcase 3)
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["back"], rk["out"], dfft.INVERSE))
    for rk in ranks:
        err = float((rk["back"] / n3 - rk["x"]).abs().max()) / 255.0
        assert err < 1e-10


def test_1024_r2c_c2r_round_trip_and_hermitian_half():
    """execR2C / execC2R at 1024^3 on 2x4 virtual ranks: uneven 513 = 129+128+128+128 split"""
    shape = (1024, 1024, 1024)
    n3 = float(np.prod(shape))
    ranks = make_world(shape, 2, 4, "double", c2c=False)
    assert [rk["plan"].getOutSize()[2] for rk in ranks[:4]] == [129, 128, 128, 128]
    run_all(ranks, lambda rk: rk["plan"].execR2C(rk["out"], rk["x"]))
    scale = abs(owner_entry(ranks, (0, 0, 0)))
    spec_rms = math.sqrt(sum(float((rk["x"] ** 2).sum()) for rk in ranks))
    for k in [(0, 0, 0), (5, 9, 512), (1000, 3, 128), (77, 600, 300)]:
        got, want = owner_entry(ranks, k), direct_dft_entry(ranks, shape, k)
        assert abs(got - want) / scale < 1e-11
        assert entry_rel(got, want, spec_rms) < forward_bound("double", n3), (k, got, want)
    run_all(ranks, lambda rk: rk["plan"].execC2R(rk["back"], rk["out"]))
    for rk in ranks:
        assert float((rk["back"] / n3 - rk["x"]).abs().max()) / 255.0 < 1e-10


@pytest.mark.parametrize("shape,P1,P2", [((1024, 1024, 1024), 2, 4), ((2048, 256, 256), 2, 2), ((256, 2048, 256), 1, 4),
                                         ((256, 256, 2048), 4, 1)])
def test_c5_fp32_axis_2048_and_1024_cube(shape, P1, P2):
    """fp32 path: C5's axis length 2048 on every axis in turn, and a 1024^3 pencil 2x4"""
    n3 = float(np.prod(shape))
    ranks = make_world(shape, P1, P2, "float")
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["out"], rk["x"], dfft.FORWARD))
    scale = abs(owner_entry(ranks, (0, 0, 0)))
    spec_rms = math.sqrt(sum(float((rk["x"].abs().to(torch.float64) ** 2).sum()) for rk in ranks))
    for k in [(0, 0, 0), (1, 2, 3), (shape[0] - 1, shape[1] // 2, shape[2] // 3)]:
        got, want = owner_entry(ranks, k), direct_dft_entry(ranks, shape, k)
        assert abs(got - want) / scale < 1e-4
        if all(k):      # off the planes through the DC point (non-negative fp32 input: parity_metric.py)
            assert entry_rel(got, want, spec_rms) < forward_bound("float", n3), (k, got, want)
    run_all(ranks, lambda rk: rk["plan"].execC2C(rk["back"], rk["out"], dfft.INVERSE))
    for rk in ranks:
        assert float((rk["back"] / n3 - rk["x"]).abs().max()) / 255.0 < 5e-5


def test_linearity_and_shift_theorem_512():
    """FFT(a x + b y) = a FFT(x) + b FFT(y); a circular shift along x multiplies by a phase"""
    shape = (512, 512, 512)
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), True, c2c=True)
    n = int(np.prod(shape))
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.view_as_complex(torch.rand((n, 2), dtype=torch.float64, device="cuda", generator=g)).reshape(shape)
    y = torch.view_as_complex(torch.rand((n, 2), dtype=torch.float64, device="cuda", generator=g)).reshape(shape)
    outs = [torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    plan.execC2C(outs[0], x, dfft.FORWARD)
    plan.execC2C(outs[1], y, dfft.FORWARD)
    z = (2.5 * x - 1.25j * y).contiguous()
    torch.cuda.synchronize()
    plan.execC2C(outs[2], z, dfft.FORWARD)
    lin = 2.5 * outs[0] - 1.25j * outs[1]
    assert float((outs[2] - lin).abs().max() / lin.abs().max()) < 1e-12
    xs = torch.roll(x, shifts=3, dims=0).contiguous()
    torch.cuda.synchronize()
    plan.execC2C(outs[1], xs, dfft.FORWARD)
    kx = torch.arange(shape[0], device="cuda", dtype=torch.float64)
    phase = torch.polar(torch.ones_like(kx), -2.0 * math.pi * 3 * kx / shape[0]).reshape(-1, 1, 1)
    want = outs[0][:n].reshape(shape) * phase
    assert float((outs[1][:n].reshape(shape) - want).abs().max() / want.abs().max()) < 1e-12
