"""ctypes front-end of the CPU oracle (oracle/dfft_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by distributedfft_amd/.  See the header of dfft_oracle.c for what is
restated and how the oracle is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc + OpenMP)."""
    src = os.path.join(_HERE, "dfft_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        sz, vp, i32, u64 = C.c_size_t, C.c_void_p, C.c_int, C.c_uint64
        L.orc_dft_naive.argtypes = [vp, vp, sz, i32]
        L.orc_fft1d_many.argtypes = [vp, sz, sz, sz, sz, i32]
        L.orc_fft3d_c2c.argtypes = [vp, sz, sz, sz, i32]
        L.orc_fft3d_r2c.argtypes = [vp, vp, sz, sz, sz]
        L.orc_fft3d_c2r.argtypes = [vp, vp, sz, sz, sz]
        L.orc_plan_create.restype = vp
        L.orc_plan_create.argtypes = [sz, sz, sz, i32, i32, i32]
        L.orc_plan_destroy.argtypes = [vp]
        L.orc_plan_in_block.argtypes = [vp, i32, vp, vp]
        L.orc_plan_out_block.argtypes = [vp, i32, vp, vp]
        L.orc_plan_domain_elems.restype = sz
        L.orc_plan_domain_elems.argtypes = [vp, i32]
        L.orc_plan_exchange_tables.argtypes = [vp, i32, i32, vp, vp, vp, vp]
        L.orc_pencil_forward.argtypes = [vp, vp, vp]
        L.orc_pencil_inverse.argtypes = [vp, vp, vp]
        L.orc_uniform255.restype = C.c_double
        L.orc_uniform255.argtypes = [u64, u64]
        L.orc_fill_block.argtypes = [vp, sz, sz, sz, sz, sz, sz, sz, sz, i32, u64]
        L.orc_derivative_coefficients.argtypes = [vp, sz, sz, sz, sz, sz, sz, sz, i32]
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        _lib = L
    return _lib


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def usable_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 cores and an
    affinity of 256, but cpu.max grants 16: 128 OpenMP threads or MPI ranks then share 16 cores' worth of time)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    return lib().orc_num_threads()


def dft_naive(x, sign=-1):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_dft_naive(_p(x), _p(out), x.size, sign)
    return out


def fft1d(x, sign=-1):
    """Batched 1-D transform along the last axis (copy)."""
    a = np.array(x, dtype=np.complex128, order="C", copy=True)
    n = a.shape[-1]
    lib().orc_fft1d_many(_p(a), n, 1, n, a.size // n, sign)
    return a


def fft3d_c2c(x, sign=-1):
    a = np.array(x, dtype=np.complex128, order="C", copy=True)
    lib().orc_fft3d_c2c(_p(a), *a.shape, sign)
    return a


def fft3d_r2c(x):
    a = np.ascontiguousarray(x, dtype=np.float64)
    Nx, Ny, Nz = a.shape
    out = np.empty((Nx, Ny, Nz // 2 + 1), dtype=np.complex128)
    lib().orc_fft3d_r2c(_p(a), _p(out), Nx, Ny, Nz)
    return out


def fft3d_c2r(X, Nz):
    a = np.array(X, dtype=np.complex128, order="C", copy=True)
    Nx, Ny, Nzc = a.shape
    assert Nzc == Nz // 2 + 1
    out = np.empty((Nx, Ny, Nz), dtype=np.float64)
    lib().orc_fft3d_c2r(_p(a), _p(out), Nx, Ny, Nz)
    return out


def fill_block(shape_global, start, size, ncomp, seed):
    """Deterministic uniform[0,255) block of the global grid (value depends on the global
    linear index only, so every decomposition sees the same grid)."""
    Nx, Ny, Nz = shape_global
    out = np.empty(tuple(size) + ((ncomp,) if ncomp > 1 else ()), dtype=np.float64)
    lib().orc_fill_block(_p(out), Ny, Nz, start[0], start[1], start[2], size[0], size[1], size[2],
                         ncomp, seed)
    if ncomp == 2:
        return out.view(np.complex128)[..., 0]
    return out


def derivative_coefficients(block, Nglobal, Nz_offset, Ny_offset, half):
    """In-place restatement of the reference's derivativeCoefficients kernel."""
    Nx, N2, N1 = block.shape
    assert block.dtype == np.complex128 and block.flags.c_contiguous
    lib().orc_derivative_coefficients(_p(block), Nglobal[0], Nglobal[1], Nglobal[2], Nz_offset,
                                      Ny_offset, N1, N2, int(half))
    return block


def testcase4_root(shape):
    """the divisor of the reference's derivativeCoefficients: sqrtf (single precision) of the int product Nx*Ny*Nz
    (tests/src/pencil/random_dist_3D.cu:117-118)"""
    return float(np.sqrt(np.float32(int(shape[0]) * int(shape[1]) * int(shape[2]))))


def testcase4_expected(shape, u):
    """what forward -> derivativeCoefficients -> unnormalised inverse yields for u = sin sin sin in the reference's arithmetic:
    -3 N^3 / sqrtf(N^3) u (the test compares it with -3 sqrt(N^3) u, random_dist_3D.cu:758-762)"""
    n3 = float(shape[0]) * shape[1] * shape[2]
    return -3.0 * n3 / testcase4_root(shape) * u


def testcase4_printed(shape):
    """(avg, max) the reference's testcase 4 prints, up to the rounding of the transforms: 3 |N^3/sqrtf(N^3) - sqrt(N^3)| times the
    mean / max of |sin sin sin| over the grid"""
    Nx, Ny, Nz = shape
    n3 = float(Nx) * Ny * Nz
    dev = 3.0 * abs(n3 / testcase4_root(shape) - np.sqrt(n3))
    sx, sy, sz = (np.abs(np.sin(2 * np.pi * np.arange(n) / n)) for n in shape)
    return dev * sx.mean() * sy.mean() * sz.mean(), dev * sx.max() * sy.max() * sz.max()


class PencilPlan:
    """Virtual-rank restatement of MPIcuFFT_Pencil_Opt1 (slab == P2 = 1)."""

    def __init__(self, Nx, Ny, Nz, P1, P2, c2c):
        self.N = (Nx, Ny, Nz)
        self.P1, self.P2, self.P = P1, P2, P1 * P2
        self.c2c = bool(c2c)
        self.Nzc = Nz if c2c else Nz // 2 + 1
        self._h = lib().orc_plan_create(Nx, Ny, Nz, P1, P2, int(c2c))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_plan_destroy(self._h)
            self._h = None

    def _block(self, fn, rank):
        s = (C.c_size_t * 3)()
        o = (C.c_size_t * 3)()
        fn(self._h, rank, s, o)
        return tuple(s), tuple(o)

    def in_block(self, rank):
        return self._block(lib().orc_plan_in_block, rank)

    def out_block(self, rank):
        return self._block(lib().orc_plan_out_block, rank)

    def domain_elems(self, rank):
        return lib().orc_plan_domain_elems(self._h, rank)

    def exchange_tables(self, rank, which):
        n = self.P2 if which == 1 else self.P1
        arrs = [(C.c_size_t * n)() for _ in range(4)]
        lib().orc_plan_exchange_tables(self._h, rank, which, *arrs)
        return [list(a) for a in arrs]

    def scatter(self, g):
        """Split a global input grid into per-rank input blocks."""
        blocks = []
        for r in range(self.P):
            s, o = self.in_block(r)
            blocks.append(np.ascontiguousarray(g[o[0]:o[0] + s[0], o[1]:o[1] + s[1], :]))
        return blocks

    def gather_out(self, outs):
        """Assemble per-rank output blocks [Nx][yo][zs] into the global [Nx][Ny][Nzc]."""
        G = np.empty((self.N[0], self.N[1], self.Nzc), dtype=np.complex128)
        for r in range(self.P):
            s, o = self.out_block(r)
            n = s[0] * s[1] * s[2]
            G[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]] = outs[r].ravel()[:n].reshape(s)
        return G

    def forward(self, in_blocks):
        dt = np.complex128 if self.c2c else np.float64
        ins = [np.ascontiguousarray(b, dtype=dt) for b in in_blocks]
        outs = [np.zeros(self.domain_elems(r), dtype=np.complex128) for r in range(self.P)]
        pin = (C.c_void_p * self.P)(*[b.ctypes.data for b in ins])
        pout = (C.c_void_p * self.P)(*[b.ctypes.data for b in outs])
        lib().orc_pencil_forward(self._h, pin, pout)
        return outs

    def inverse(self, spec_blocks):
        """spec_blocks[r]: flat/shape [Nx][yo][zs] spectrum; returns [xs][ys][Nz] blocks."""
        dt = np.complex128 if self.c2c else np.float64
        ins = []
        for r in range(self.P):
            buf = np.zeros(self.domain_elems(r), dtype=np.complex128)
            v = np.asarray(spec_blocks[r], dtype=np.complex128).ravel()
            s, _ = self.out_block(r)
            n = s[0] * s[1] * s[2]
            buf[:n] = v[:n]
            ins.append(buf)
        outs = [np.zeros(self.in_block(r)[0], dtype=dt) for r in range(self.P)]
        pin = (C.c_void_p * self.P)(*[b.ctypes.data for b in ins])
        pout = (C.c_void_p * self.P)(*[b.ctypes.data for b in outs])
        lib().orc_pencil_inverse(self._h, pin, pout)
        return outs
