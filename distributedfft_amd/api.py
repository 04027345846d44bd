"""Host-side mirror of the reference's plan/execute interface for the hot path.

Same names, argument meaning and error behaviour as the reference's C++ classes
(paths relative to the reference repository):

    GlobalSize, Partition, Slab_Partition, Pencil_Partition, Configurations
                                                         include/params.hpp:24-93
    MPIcuFFT<T> (abstract)                               include/mpicufft.hpp:55-105
    MPIcuFFT_Slab / MPIcuFFT_Slab_Opt1                   include/mpicufft_slab.hpp, _slab_opt1.hpp
    MPIcuFFT_Pencil / MPIcuFFT_Pencil_Opt1               include/mpicufft_pencil.hpp, _pencil_opt1.hpp

Everything here is plumbing over the C ABI (include/dfft_c.h): device pointers in, device
pointers out.  torch is used only to own device memory and for torch.distributed.
"""
import ctypes as C
from dataclasses import dataclass

from . import _lib
from ._lib import Config, DfftError, check, lib

# include/params.hpp:83-84
Peer2Peer, All2All = 0, 1
Sync, Streams, MPI_Type = 0, 1, 2
FORWARD, INVERSE = -1, 1


class GlobalSize:
    """include/params.hpp:24-37"""

    def __init__(self, Nx, Ny, Nz):
        self.Nx, self.Ny, self.Nz = int(Nx), int(Ny), int(Nz)
        self.Nz_out = self.Nz // 2 + 1


class Partition:
    """include/params.hpp:39-42"""

    def __init__(self, P1=1, P2=1):
        self.P1, self.P2 = int(P1), int(P2)


class Slab_Partition(Partition):
    """include/params.hpp:44-49"""

    def __init__(self, P1):
        super().__init__(P1, 1)


class Pencil_Partition(Partition):
    """include/params.hpp:51-56"""

    def __init__(self, P1, P2):
        super().__init__(P1, P2)


class Partition_Dimensions:
    """include/params.hpp:58-81"""

    def __init__(self):
        self.size_x, self.size_y, self.size_z = [], [], []
        self.start_x, self.start_y, self.start_z = [], [], []

    def computeOffsets(self):
        for n in "xyz":
            off, starts = 0, []
            for v in getattr(self, "size_" + n):
                starts.append(off)
                off += v
            setattr(self, "start_" + n, starts)


@dataclass
class Configurations:
    """include/params.hpp:85-93"""
    cuda_aware: bool = True
    warmup_rounds: int = 0
    comm_method: int = All2All
    send_method: int = Sync
    benchmark_dir: str = "../benchmarks"
    comm_method2: int = All2All
    send_method2: int = Sync

    def _c(self):
        return Config(int(self.cuda_aware), self.warmup_rounds, self.comm_method, self.send_method,
                      self.comm_method2, self.send_method2)


def _ptr(x):
    """device pointer of a torch tensor / int / None"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"expected a device tensor or an integer address, got {type(x)}")


class DeviceBuffer:
    """Device memory from dfft_malloc / dfft_tune_placement (include/dfft_c.h): an address and a size.  Plans take it
    wherever they take a tensor; `tensor(dtype)` is a zero-copy torch view (through __cuda_array_interface__) for the
    callers that fill or check the data with torch ops.  Freed with dfft_free when the object dies."""

    def __init__(self, address, nbytes, owned=True):
        self.address, self.nbytes, self._owned = int(address), int(nbytes), owned

    CHUNK_DEFAULT = 2 ** (8 * C.sizeof(C.c_size_t)) - 1      # DFFT_CHUNK_DEFAULT of include/dfft_c.h

    @classmethod
    def alloc(cls, nbytes, chunk_mib=None):
        """chunk_mib: None = the library's default backing (virtual-memory API, 1 GiB physical chunks; what the library uses for
        the work areas it owns), 0 = plain hipMalloc, k = physical chunks of k MiB"""
        h = C.c_void_p()
        check(lib().dfft_malloc(int(nbytes), cls.CHUNK_DEFAULT if chunk_mib is None else int(chunk_mib), C.byref(h)))
        return cls(h.value, nbytes)

    def data_ptr(self):
        return self.address

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.address, False), "version": 2, "strides": None}

    def tensor(self, dtype):
        import torch
        t = torch.as_tensor(self, device="cuda")
        if t.data_ptr() != self.address:
            raise DfftError("torch copied the buffer instead of viewing it")
        t = t.view(dtype)
        t._dfft_owner = self      # the view keeps the allocation alive
        return t

    def free(self):
        if self._owned and self.address:
            lib().dfft_free(C.c_void_p(self.address))
        self.address = 0

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass


def last_placement_info():
    """what the last placement-aware allocation of this process did (dfft_last_placement_info): K, candidates, probe rates, seconds"""
    import json
    buf = C.create_string_buffer(1024)
    check(lib().dfft_last_placement_info(buf, 1024))
    return json.loads(buf.value.decode())


class Comm:
    """Communicator handed to the plans in place of MPI_Comm (src/mpicufft.cpp:42-51)."""

    def __init__(self, handle, nranks, rank=None, keep=None):
        self._h, self.nranks, self.rank, self._keep = handle, nranks, rank, keep

    @classmethod
    def local(cls, nranks):
        """nranks virtual ranks in this process on the current device (one host thread each)."""
        h = C.c_void_p()
        check(lib().dfft_comm_create_local(nranks, C.byref(h)))
        return cls(h, nranks)

    @classmethod
    def rccl(cls, unique_id, nranks, rank):
        """one process per GPU over RCCL/xGMI; unique_id = bytes from rccl_unique_id() of rank 0"""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        h = C.c_void_p()
        check(lib().dfft_comm_create_rccl(buf, nranks, rank, C.byref(h)))
        return cls(h, nranks, rank)

    @classmethod
    def callback(cls, nranks, rank, fn):
        """fn(send_ptr, scounts, sdispls, recv_ptr, rcounts, rdispls, group, me, stream) -> None"""

        def tramp(user, send, sc, sd, recv, rc, rd, group, ng, me, stream):
            try:
                fn(send, [sc[i] for i in range(ng)], [sd[i] for i in range(ng)], recv,
                   [rc[i] for i in range(ng)], [rd[i] for i in range(ng)], [group[i] for i in range(ng)], me,
                   stream)
                return 0
            except Exception:  # surfaced as a library error with the traceback printed
                import traceback
                traceback.print_exc()
                return 1

        cfn = _lib.ALLTOALLV_FN(tramp)
        h = C.c_void_p()
        check(lib().dfft_comm_create_callback(nranks, rank, cfn, None, C.byref(h)))
        return cls(h, nranks, rank, keep=cfn)

    def setListCallback(self, fn):
        """optional second callback of a callback communicator (dfft_comm_set_list_callback): a point-to-point schedule in one call,
        fn(sends, recvs, stream) with sends / recvs = [(peer, ptr, nbytes), ...] in matching order at both ends of every link"""

        def tramp(user, ns, sp, sptr, sby, nr, rp, rptr, rby, stream):
            try:
                fn([(sp[i], sptr[i] or 0, sby[i]) for i in range(ns)], [(rp[i], rptr[i] or 0, rby[i]) for i in range(nr)], stream)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        cfn = _lib.SENDRECV_LIST_FN(tramp)
        check(lib().dfft_comm_set_list_callback(self._h, cfn, None))
        self._keep = (self._keep, cfn)

    def counters(self):
        """what went through the transport since the communicator was made (dfft_comm_get_counter): all-to-all-v operations, native
        point-to-point schedules (one per hop of a relayed exchange), relayed exchanges, table gathers of the relay, and its one-word
        agreements (one per exchange table and stream at first use: every rank could set up its staging)"""
        out = {}
        for k in ("alltoallv", "list", "relayed", "relay_meta", "relay_agree"):
            v = C.c_long(0)
            check(lib().dfft_comm_get_counter(self._h, k.encode(), C.byref(v)))
            out[k] = v.value
        return out

    def getCounter(self, name):
        """one counter (dfft_comm_get_counter); "layered" is non-zero while the library runs a schedule as all-to-all-v layers"""
        v = C.c_long(0)
        check(lib().dfft_comm_get_counter(self._h, name.encode(), C.byref(v)))
        return v.value

    @staticmethod
    def rccl_unique_id():
        buf = C.create_string_buffer(128)
        check(lib().dfft_rccl_unique_id(buf))
        return buf.raw

    def info(self):
        """(nranks, transport_nranks): the second is ncclCommCount for the RCCL transport, 0 otherwise"""
        a, b = C.c_int(0), C.c_int(0)
        check(lib().dfft_comm_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def setOption(self, key, value):
        """transport knobs (include/dfft_c.h: dfft_comm_set_option); 'dup_channel' = 1 is collective over all ranks"""
        check(lib().dfft_comm_set_option(self._h, key.encode(), int(value)))

    def alltoallv(self, myrank, send, scounts, sdispls, recv, rcounts, rdispls, group, me, stream=None):
        """the transport's all-to-all-v by itself (dfft_comm_alltoallv): byte counts / displacements, stream-ordered"""
        n = len(group)
        arr = lambda v: (C.c_size_t * n)(*[int(x) for x in v])      # noqa: E731
        check(lib().dfft_comm_alltoallv(self._h, int(myrank), _ptr(send), arr(scounts), arr(sdispls), _ptr(recv), arr(rcounts), arr(rdispls),
                                        (C.c_int * n)(*[int(g) for g in group]), n, int(me), C.c_void_p(stream or 0)))

    def sendrecvList(self, myrank, sends, recvs, nlayers, stream=None):
        """the transport's point-to-point schedule by itself (dfft_comm_sendrecv_list): sends / recvs = [(peer, layer, buffer or address,
        nbytes), ...]; one grouped operation on the RCCL / local-world / list-callback transports"""
        def pack(lst):
            n = len(lst)
            return (n, (C.c_int * n)(*[int(p) for p, _, _, _ in lst]), (C.c_int * n)(*[int(l) for _, l, _, _ in lst]),
                    (C.c_void_p * n)(*[b if isinstance(b, int) else _ptr(b).value for _, _, b, _ in lst]),
                    (C.c_size_t * n)(*[int(nb) for _, _, _, nb in lst]))
        ns, sp, sl, sptr, sb = pack(sends)
        nr, rp, rl, rptr, rb = pack(recvs)
        check(lib().dfft_comm_sendrecv_list(self._h, int(myrank), ns, sp, sl, sptr, sb, nr, rp, rl, rptr, rb, int(nlayers), C.c_void_p(stream or 0)))

    def destroy(self):
        if self._h:
            lib().dfft_comm_destroy(self._h)
            self._h = None


class MPIcuFFT:
    """include/mpicufft.hpp:55-105.  precision: 'double' | 'float' (the template argument T)."""
    _kind = 3

    def __init__(self, config=None, comm=None, max_world_size=-1, precision="double", rank=0):
        self.config = config or Configurations()
        self.comm = comm
        self.precision = {"double": 1, "float": 0, "f64": 1, "f32": 0}[precision]
        cfg = self.config._c()
        self._h = C.c_void_p()
        check(lib().dfft_plan_create(C.byref(self._h), self._kind, self.precision, C.byref(cfg),
                                     comm._h if comm else None,
                                     rank if (comm is None or comm.rank is None) else comm.rank, max_world_size))
        self.c2c = False
        self._work = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().dfft_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- plan ---------------------------------------------------------------------------
    def initFFT(self, global_size, partition=None, allocate=True, c2c=False):
        """initFFT(GlobalSize*, Partition*, bool allocate).  c2c=True plans a complex
        transform (Nz_out = Nz) for execC2C instead of R2C/C2R."""
        if global_size is None:
            raise DfftError("GlobalSize or Partition not initialized!")
        if partition is None:
            partition = Partition(lib().dfft_world_size(self._h), 1)
        self.global_size, self.partition, self.c2c = global_size, partition, bool(c2c)
        check(lib().dfft_init(self._h, global_size.Nx, global_size.Ny, global_size.Nz, partition.P1,
                              partition.P2, int(c2c), int(allocate)))

    def setWorkArea(self, device=None, host=None):
        self._work = device
        check(lib().dfft_set_work_area(self._h, _ptr(device), _ptr(host)))

    def tunePlacement(self, in_, tries=4, want_back=True):
        """dfft_tune_placement: tries several physical backings for the library-owned work area, a new `out` buffer and
        (want_back) a new buffer for the inverse transform's output, and keeps for each the one this plan's own passes run
        fastest on.  `in_` must hold a valid input block.  Returns (out, back or None, [ms of every trial])."""
        o, b = C.c_void_p(), C.c_void_p()
        rep = (C.c_float * (3 * max(1, int(tries)) + 32))()
        n = C.c_int(0)
        check(lib().dfft_tune_placement(self._h, _ptr(in_), int(tries), C.byref(o), C.byref(b) if want_back else None, rep, len(rep), C.byref(n)))
        isz = self.getInSize()
        esz = 16 if self.precision == 1 else 8
        in_bytes = isz[0] * isz[1] * isz[2] * (esz if self.c2c else esz // 2)
        out = DeviceBuffer(o.value, self.getDomainSize())
        back = DeviceBuffer(b.value, in_bytes) if want_back else None
        return out, back, [float(rep[i]) for i in range(n.value)]

    def tuneVariants(self, in_, out, back=None):
        """dfft_tune_variants: every pass tries the four workgroup orders and every kernel configuration of its length on these
        buffers and keeps what runs faster here.  Returns the measured FFT ms (forward + inverse) of every trial: the plan as built,
        the four order settings, the chosen orders, one entry per configuration number, the final choice."""
        rep = (C.c_float * 32)()
        n = C.c_int(0)
        check(lib().dfft_tune_variants(self._h, _ptr(in_), _ptr(out), _ptr(back), rep, len(rep), C.byref(n)))
        return [float(rep[i]) for i in range(n.value)]

    def setPipelineChunks(self, chunks):
        """pipeline depth of the exchanges (before initFFT); 1 = no overlap, 0 = default"""
        check(lib().dfft_set_pipeline_chunks(self._h, int(chunks)))

    def getPipelineChunks(self):
        return lib().dfft_get_pipeline_chunks(self._h)

    def setOption(self, key, value):
        """named tuning knob (dfft_set_option, include/dfft_c.h); effective at the next initFFT"""
        check(lib().dfft_set_option(self._h, key.encode(), int(value)))

    def getOption(self, key):
        return lib().dfft_get_option(self._h, key.encode())

    def setStream(self, stream):
        """HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream)"""
        check(lib().dfft_set_stream(self._h, C.c_void_p(int(stream))))

    # -- execute ------------------------------------------------------------------------
    def execR2C(self, out, in_, d=3):
        """execR2C(out, in) and the pencil classes' execR2C(out, in, d) (d = 1, 2: partial)"""
        if d == 3:
            check(lib().dfft_exec_r2c(self._h, _ptr(out), _ptr(in_)))
        else:
            if self.c2c:
                raise DfftError("plan was initialised for C2C")
            check(lib().dfft_exec_dim(self._h, _ptr(out), _ptr(in_), FORWARD, d))

    def execC2R(self, out, in_, d=3):
        if d == 3:
            check(lib().dfft_exec_c2r(self._h, _ptr(out), _ptr(in_)))
        else:
            if self.c2c:
                raise DfftError("plan was initialised for C2C")
            check(lib().dfft_exec_dim(self._h, _ptr(out), _ptr(in_), INVERSE, d))

    def execC2C(self, out, in_, direction=FORWARD, sync=True, d=3):
        if d != 3:
            check(lib().dfft_exec_dim(self._h, _ptr(out), _ptr(in_), direction, d))
            return
        f = lib().dfft_exec_c2c if sync else lib().dfft_enqueue_c2c
        check(f(self._h, _ptr(out), _ptr(in_), direction))

    def exchange(self, which, direction, sendbuf, recvbuf):
        """only the all-to-all of exchange `which` (1 row group / 2 column group)"""
        check(lib().dfft_exchange(self._h, which, direction, _ptr(sendbuf), _ptr(recvbuf)))

    # -- getters ------------------------------------------------------------------------
    def _get3(self, fn):
        a = (C.c_size_t * 3)()
        check(fn(self._h, a))
        return tuple(a)

    def getInSize(self):
        return self._get3(lib().dfft_get_in_size)

    def getInStart(self):
        return self._get3(lib().dfft_get_in_start)

    def getOutSize(self):
        return self._get3(lib().dfft_get_out_size)

    def getOutStart(self):
        return self._get3(lib().dfft_get_out_start)

    def getPassChoices(self):
        """{pass: (variant, order, addr64)} for fz fy fx ix iy iz: what every pass runs with right now (dfft_get_pass_choices) -- the
        plan's rules, or what tuneVariants / tunePlacement kept"""
        v, o, a = (C.c_int * 6)(), (C.c_int * 6)(), (C.c_int * 6)()
        check(lib().dfft_get_pass_choices(self._h, v, o, a))
        return {n: (v[i], o[i], a[i]) for i, n in enumerate(("fz", "fy", "fx", "ix", "iy", "iz"))}

    def getOutStrides(self):
        """element strides of the spectrum block along (kx, ky_local, kz_local) (dfft_get_out_strides): (yo*zs, zs, 1) for the
        reference's [Nx][yo][zs], (1, zs*Nx, Nx) with setOption("spectral_layout", 1)"""
        return self._get3(lib().dfft_get_out_strides)

    def spectrumView(self, out):
        """the spectrum block held by the torch tensor `out` as a (Nx, yo, zs) view, whatever the plan's spectral layout -- pointwise
        work on it (a Laplacian, a filter: the reference's testcase 4) is layout-independent"""
        n, st = self.getOutSize(), self.getOutStrides()
        return out.reshape(-1).as_strided(tuple(n), tuple(st))

    def getPartitionDimensions(self):
        """(input_dim, transposed_dim, output_dim) of include/mpicufft_pencil.hpp:112-116, each a
        Partition_Dimensions with size_x/y/z and start_x/y/z lists (include/params.hpp:58-81)"""
        dims = []
        for which in range(3):
            d = Partition_Dimensions()
            for axis, name in enumerate("xyz"):
                n = C.c_size_t(0)
                check(lib().dfft_get_partition_dimensions(self._h, which, axis, None, None, 0, C.byref(n)))
                sz, st = (C.c_size_t * n.value)(), (C.c_size_t * n.value)()
                check(lib().dfft_get_partition_dimensions(self._h, which, axis, sz, st, n.value, C.byref(n)))
                setattr(d, "size_" + name, list(sz))
                setattr(d, "start_" + name, list(st))
            dims.append(d)
        return tuple(dims)

    def getDomainSize(self):
        return lib().dfft_domain_size(self._h)

    def getWorkSizeDevice(self):
        return lib().dfft_work_size_device(self._h)

    def getWorkSizeHost(self):
        return lib().dfft_work_size_host(self._h)

    def getWorkAreaDevice(self):
        return lib().dfft_work_area_device(self._h)

    def getRank(self):
        return lib().dfft_rank(self._h)

    def getWorldSize(self):
        return lib().dfft_world_size(self._h)

    def getExchangeTables(self, which):
        n = self.partition.P2 if which == 1 else self.partition.P1
        arrs = [(C.c_size_t * n)() for _ in range(4)]
        check(lib().dfft_get_exchange_tables(self._h, which, *arrs))
        return [list(a) for a in arrs]

    def getPipelineTables(self, direction, which, chunk):
        n = self.partition.P2 if which == 1 else self.partition.P1
        arrs = [(C.c_size_t * n)() for _ in range(4)]
        check(lib().dfft_get_pipeline_tables(self._h, direction, which, chunk, *arrs))
        return [list(a) for a in arrs]

    def debugPass(self, name, index=0):
        """descriptor of one axis pass (dfft_debug_get_pass); None if the plan has no such launch"""
        from ._lib import PassDesc
        d = PassDesc()
        if lib().dfft_debug_get_pass(self._h, name.encode(), int(index), C.byref(d)) != 0:
            return None
        return d

    def debugPointTable(self, name, index=0, store=False):
        """[(base, ln, aux)] per point of a segmented side (dfft_debug_get_point_table)"""
        n = C.c_size_t(0)
        check(lib().dfft_debug_get_point_table(self._h, name.encode(), int(index), int(store), None, None, None, 0, C.byref(n)))
        base, ln, aux = (C.c_uint64 * n.value)(), (C.c_uint32 * n.value)(), (C.c_uint32 * n.value)()
        check(lib().dfft_debug_get_point_table(self._h, name.encode(), int(index), int(store), base, ln, aux, n.value, C.byref(n)))
        return list(zip(base, ln, aux))

    def getTileLines(self):
        return lib().dfft_tile_lines(self._h)

    def enablePhaseTiming(self, on=True):
        check(lib().dfft_enable_phase_timing(self._h, int(on)))

    def getPhaseTimes(self, direction=FORWARD):
        ms = (C.c_float * 8)()
        n = lib().dfft_get_phase_times(self._h, ms, 8)
        return [(lib().dfft_phase_name(i, direction).decode(), ms[i]) for i in range(n)]


class MPIcuFFT_Slab(MPIcuFFT):
    _kind = 0


class MPIcuFFT_Slab_Opt1(MPIcuFFT_Slab):
    """include/mpicufft_slab_opt1.hpp:70 (derives from MPIcuFFT_Slab)"""
    _kind = 1


class MPIcuFFT_Slab_Z_Then_YX(MPIcuFFT):
    """include/mpicufft_slab_z_then_yx.hpp: input split along x, output [Nx][Ny][Nzc/P] split along z"""
    _kind = 4


class MPIcuFFT_Slab_Z_Then_YX_Opt1(MPIcuFFT_Slab_Z_Then_YX):
    _kind = 5


class MPIcuFFT_Slab_Y_Then_ZX(MPIcuFFT):
    """include/mpicufft_slab_y_then_zx.hpp: forward only; R2C along y, output [Nx][(Ny/2+1)/P][Nz]"""
    _kind = 6


class MPIcuFFT_Pencil(MPIcuFFT):
    _kind = 2


class MPIcuFFT_Pencil_Opt1(MPIcuFFT_Pencil):
    """include/mpicufft_pencil_opt1.hpp:23 (derives from MPIcuFFT_Pencil)"""
    _kind = 3


def fft1d_batched(out, in_, N, batch, direction=FORWARD, precision="double", stream=0, variant=0, debug=0):
    """one axis pass on natural lines [batch][N] (kernel-level entry point)"""
    prec = {"double": 1, "float": 0}[precision]
    check(lib().dfft_fft1d_batched_ex(prec, N, batch, _ptr(out), _ptr(in_), direction, C.c_void_p(int(stream)),
                                      int(variant), int(debug)))


def axis_plan_info(N, precision="double", two_level=0):
    """how an axis of N points is transformed: {"kind": "native" | "bluestein" | "two_level" | "long_bluestein", "M": inner length,
    "levels": [(N1, M1, bluestein1), (N2, M2, bluestein2)]}; None if the length has no plan"""
    prec = {"double": 1, "float": 0}[precision]
    info = (C.c_size_t * 8)()
    if lib().dfft_axis_plan_info(prec, N, int(two_level), info) != 0:
        return None
    kind = ("native", "bluestein", "two_level", "long_bluestein")[info[0]]
    # long_bluestein: Bluestein's algorithm on M padded points, whose M-point transforms run in the two levels listed
    levels = [(info[2], info[3], bool(info[4])), (info[5], info[6], bool(info[7]))] if info[0] >= 2 else []
    return {"kind": kind, "M": info[1], "levels": levels}


def kernel_info(N, precision="double"):
    prec = {"double": 1, "float": 0}[precision]
    vals = [C.c_int() for _ in range(4)]
    rc = lib().dfft_kernel_info(prec, N, *[C.byref(v) for v in vals])
    if rc != 0:
        return None
    return dict(zip(("threads", "lds_bytes", "points_per_thread", "lines_per_workgroup"), (v.value for v in vals)))
