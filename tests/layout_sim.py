"""CPU model of the product's pass descriptors (test infrastructure, not a product path).

The HIP kernels are addressed by `PassArgs` descriptors that the host builds per plan
(distributedfft_amd/csrc/dfft.hip: build_pipeline, build_pipeline_zyx, build_pipeline_yzx).  This
module reads those descriptors through the C ABI's introspection getters
(dfft_debug_get_pass / dfft_debug_get_point_table, host only) and executes the documented address
forms (fft_pass.hip.h: LoadKind / StoreKind, SegEntry) with numpy transforms, rank by rank, moving
blocks between virtual ranks with the plan's own chunked exchange tables.  It checks what a GPU run
cannot isolate: that every descriptor, segment table and per-point table of a plan describes a
consistent data flow that ends in the reference's output layout -- without a GPU.

Buffer routing mirrors enqueue_forward / enqueue_inverse / enqueue_*_zyx / enqueue_forward_yzx /
enqueue_partial_* in dfft.hip and must be kept in step with them.
"""
import numpy as np

import distributedfft_amd as dfft

LINES, TILED, KMAJOR = 0, 1, 2
S_LINES, S_KMAJOR, S_SAME, S_TRANSPOSE = 0, 1, 2, 3
ESZ = {"double": 16, "float": 8}      # bytes per complex element; a real element is half of it


def _seg(starts, lens, bases, n):
    s = 0
    for q in range(1, len(starts)):
        if n >= starts[q]:
            s = q
    return starts[s], lens[s], bases[s]


class Pass:
    """one launch descriptor + its per-point tables"""

    def __init__(self, plan, name, index=0):
        self.esz = 16 if plan.precision == 1 else 8
        self.d = plan.debugPass(name, index)
        assert self.d is not None, (name, index)
        d = self.d
        self.TL = plan.getTileLines()
        self.l = ([int(v) for v in d.lstart[:d.lnseg]], [int(v) for v in d.llen[:d.lnseg]], [int(v) for v in d.lbase[:d.lnseg]])
        self.s = ([int(v) for v in d.sstart[:d.snseg]], [int(v) for v in d.slen[:d.snseg]], [int(v) for v in d.sbase[:d.snseg]])
        self.ltab = plan.debugPointTable(name, index, False) if d.load_kind == TILED and d.lnseg >= 1 else None
        self.stab = plan.debugPointTable(name, index, True) if d.store_kind in (S_SAME, S_TRANSPOSE) and d.snseg >= 1 else None

    def tile(self, line):
        b, l = divmod(line, self.TL)
        return b, l, min(self.TL, self.d.LB - b * self.TL)

    def load_offset(self, a, line, n, NP):
        d, TL = self.d, self.TL
        b, l, tw = self.tile(line)
        if d.load_kind == LINES:
            return (a * d.AS_in + line * d.KS_in if d.KS_in else (a * d.LB + line) * NP) + n
        if d.load_kind == KMAJOR:
            return n * d.KS_in + a * d.AS_in + line
        s0, ln, bs = _seg(*self.l, n)
        if d.IA:                                   # one segment with explicit (padded) strides: closed form only
            assert d.lnseg == 1
            return bs + a * d.IA + b * d.IB + (n - s0) * tw + l
        off = bs + a * ln * d.LB + b * TL * ln + (n - s0) * tw + l
        base, tln, aux = self.ltab[n]            # the per-point table must say the same
        assert off == base + tln * (a * d.LB + b * TL) + aux * tw + l
        return off

    def store_offset(self, a, line, k, NP):
        d, TL = self.d, self.TL
        b, l, tw = self.tile(line)
        if d.store_kind == S_LINES:
            return (a * d.AS_out + line * d.KS_out if d.KS_out else (a * d.LB + line) * NP) + k
        if d.store_kind == S_KMAJOR:
            return k * d.KS_out + a * d.AS_out + line
        s0, ln, bs = _seg(*self.s, k)
        base, tln, aux = self.stab[k]
        if d.store_kind == S_SAME:
            sk, sb = (d.SK or d.LB * d.LA), (d.SB or TL * d.LA)
            off = bs + (k - s0) * sk + b * sb + a * tw + l
            assert off == base + b * sb + a * tw + l
            return off
        T2 = 1 << d.T2shift
        kt, kr = (k - s0) >> d.T2shift, (k - s0) & (T2 - 1)
        tw2 = min(T2, ln - kt * T2)
        off = bs + a * ln * d.LB + kt * T2 * d.LB + line * tw2 + kr
        assert off == base + tln * (a * d.LB) + line * aux
        return off

    def run(self, src, dst, N, mode="c2c"):
        """mode: c2c (N complex points), r2c (N reals in -> N//2+1 out), c2r (N//2+1 in -> N reals out).
        src / dst are flat numpy arrays of the element type of that side."""
        d = self.d
        nin = N // 2 + 1 if mode == "c2r" else N
        nout = N // 2 + 1 if mode == "r2c" else N
        src = src[d.in_off // (self.esz // 2 if mode == "r2c" else self.esz):]
        dst = dst[d.out_off // (self.esz // 2 if mode == "c2r" else self.esz):]
        for a in range(d.na):
            for line in range(d.LB):
                x = np.array([src[self.load_offset(a, line, n, nin)] for n in range(nin)])
                if mode == "r2c":
                    y = np.fft.rfft(x.real, N)
                elif mode == "c2r":
                    y = np.fft.irfft(x, N) * N
                else:
                    y = np.fft.ifft(x) * N if d.swap else np.fft.fft(x)
                for k in range(nout):
                    dst[self.store_offset(a, line, k, nout)] = y[k]


class World:
    """P virtual ranks of one plan class on a small fp64 grid"""

    def __init__(self, cls, shape, P1, P2, c2c, chunks=None, precision="double", options=None):
        self.shape, self.P1, self.P2, self.c2c = shape, P1, P2, c2c
        self.esz = ESZ[precision]
        self.P = P1 * P2
        comm = dfft.Comm.local(self.P) if self.P > 1 else None
        self.plans = []
        for r in range(self.P):
            pl = cls(dfft.Configurations(), comm, precision=precision, rank=r)
            if chunks is not None:
                pl.setPipelineChunks(chunks)
            for k, v in (options or {}).items():
                pl.setOption(k, v)
            pl.initFFT(dfft.GlobalSize(*shape), dfft.Partition(P1, P2), allocate=False, c2c=c2c)
            self.plans.append(pl)
        self.C = self.plans[0].getPipelineChunks()
        self.nel = [pl.getDomainSize() // self.esz for pl in self.plans]
        # a work-area slice may be larger than the domain (padded private layouts of the single-rank z, x, y order)
        self.wel = [max(pl.getDomainSize(), pl.getWorkSizeDevice() // max(1, (self.P1 > 1) + (self.P2 > 1) + 1)) // self.esz for pl in self.plans]
        self.single = self.P == 1 and self.plans[0].debugPass("sz") is not None
        # option spectral_layout: the spectrum stays x-contiguous and a single rank's inverse runs the mirrored pass order (dfft_init)
        self.spectral = bool((options or {}).get("spectral_layout", 0))

    def buffers(self, n=3):
        return [[np.full(self.wel[r], np.nan + 0j, dtype=np.complex128) for _ in range(n)] for r in range(self.P)]

    def group(self, r, which):
        i, j = divmod(r, self.P2)
        return ([i * self.P2 + q for q in range(self.P2)], j) if which == 1 else ([q * self.P2 + j for q in range(self.P1)], i)

    def exchange(self, direction, which, c, send, recv):
        """all ranks: chunk c of exchange `which`; send/recv are per-rank flat complex arrays"""
        tabs = [pl.getPipelineTables(direction, which, c) for pl in self.plans]
        for r in range(self.P):
            grp, me = self.group(r, which)
            _, _, rc, rd = tabs[r]
            for q, peer in enumerate(grp):
                psc, psd, _, _ = tabs[peer]
                assert psc[me] == rc[q]
                n, e = rc[q] // self.esz, self.esz
                recv[r][rd[q] // e: rd[q] // e + n] = send[peer][psd[me] // e: psd[me] // e + n]

    # -- chains ------------------------------------------------------------------------------
    def forward(self, ins, kind="default"):
        Nx, Ny, Nz = self.shape
        P1, P2, C, pls = self.P1, self.P2, self.C, self.plans
        outs = [np.full(n, np.nan + 0j, dtype=np.complex128) for n in self.nel]
        W = self.buffers()
        zmode = "c2c" if self.c2c else "r2c"
        if kind == "default" and self.single:
            # one rank, complex: pass order z, x, y through L1 (out) and the padded L2 (work), enqueue_single
            pl = pls[0]
            Pass(pl, "sz").run(ins[0], outs[0], Nz)
            Pass(pl, "sx").run(outs[0], W[0][0], Nx)
            Pass(pl, "sy").run(W[0][0], outs[0], Ny)
            return outs
        if kind == "default":
            ysrc = [W[r][0] if P2 > 1 else outs[r] for r in range(self.P)]
            nxt = 1 if P2 > 1 else 0
            ydst = [W[r][nxt] for r in range(self.P)]
            xsrc = [W[r][nxt + 1] if P1 > 1 else ydst[r] for r in range(self.P)]
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "fz", c).run(ins[r], outs[r], Nz, zmode)
                if P2 > 1:
                    self.exchange(dfft.FORWARD, 1, c, outs, ysrc)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "fy", c).run(ysrc[r], ydst[r], Ny)
                if P1 > 1:
                    self.exchange(dfft.FORWARD, 2, c, ydst, xsrc)
            for r, pl in enumerate(pls):
                Pass(pl, "fx").run(xsrc[r], outs[r], Nx)
        elif kind == "zyx":
            P = P1
            ysrc = [W[r][0] if P > 1 else outs[r] for r in range(self.P)]
            ydst = [W[r][1] if P > 1 else W[r][0] for r in range(self.P)]
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "fz", c).run(ins[r], outs[r], Nz, zmode)
                if P > 1:
                    self.exchange(dfft.FORWARD, 2, c, outs, ysrc)
            for c in range(C):
                for r, pl in enumerate(pls):
                    for q in range(P):
                        Pass(pl, "zy", c * P + q).run(ysrc[r], ydst[r], Ny)
            for r, pl in enumerate(pls):
                Pass(pl, "fx").run(ydst[r], outs[r], Nx)
        elif kind == "yzx":
            P = P1
            xsrc = [W[r][0] if P > 1 else outs[r] for r in range(self.P)]
            xdst = [W[r][1] if P > 1 else W[r][0] for r in range(self.P)]
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "fy", c).run(ins[r], outs[r], Ny, "c2c" if self.c2c else "r2c")
                if P > 1:
                    self.exchange(dfft.FORWARD, 2, c, outs, xsrc)
            for r, pl in enumerate(pls):
                Pass(pl, "fx").run(xsrc[r], xdst[r], Nx)
                Pass(pl, "yz").run(xdst[r], outs[r], Nz)
        return outs

    def inverse(self, spec, kind="default"):
        """spec: per-rank flat complex arrays in the output layout (destroyed); returns per-rank
        flat arrays in the input layout (real for R2C plans)"""
        Nx, Ny, Nz = self.shape
        P1, P2, C, pls = self.P1, self.P2, self.C, self.plans
        zmode = "c2c" if self.c2c else "c2r"
        nin = [int(np.prod(pl.getInSize())) for pl in pls]
        outs = [np.full(n, np.nan, dtype=np.complex128 if self.c2c else np.float64) for n in nin]
        W = self.buffers(2)
        if kind == "default" and self.single:
            pl = pls[0]
            for name, src, dst, N in (("sz", spec[0], outs[0], Nz), ("sx", outs[0], W[0][0], Nx), ("sy", W[0][0], outs[0], Ny)):
                p = Pass(pl, name)
                p.d.swap = 1
                p.run(src, dst, N)
            return outs
        if kind == "default" and self.P == 1 and self.c2c and not self.spectral:
            # single rank, complex: forward pass order with conjugation (enqueue_inverse fast path)
            pl = pls[0]
            for name, src, dst, N in (("fz", spec[0], W[0][0], Nz), ("fy", W[0][0], spec[0], Ny), ("fx", spec[0], outs[0], Nx)):
                for c in range(C if name != "fx" else 1):
                    p = Pass(pl, name, c)
                    p.d.swap = 1
                    p.run(src, dst, N)
            return outs
        if kind == "default":
            xdst = [W[r][0] for r in range(self.P)]
            ysrc = [W[r][1] if P1 > 1 else W[r][0] for r in range(self.P)]
            zsrc = [(W[r][0] if P1 > 1 else W[r][1]) if P2 > 1 else spec[r] for r in range(self.P)]
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "ix", c).run(spec[r], xdst[r], Nx)
                if P1 > 1:
                    self.exchange(dfft.INVERSE, 2, c, xdst, ysrc)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "iy", c).run(ysrc[r], spec[r], Ny)
                if P2 > 1:
                    self.exchange(dfft.INVERSE, 1, c, spec, zsrc)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "iz", c).run(zsrc[r], outs[r], Nz, zmode)
        elif kind == "zyx":
            P = P1
            zsrc = [W[r][1] if P > 1 else spec[r] for r in range(self.P)]
            for r, pl in enumerate(pls):
                Pass(pl, "zix").run(spec[r], W[r][0], Nx)
            for c in range(C):
                for r, pl in enumerate(pls):
                    for q in range(P):
                        Pass(pl, "ziy", c * P + q).run(W[r][0], spec[r], Ny)
                if P > 1:
                    self.exchange(dfft.INVERSE, 2, c, spec, zsrc)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "iz", c).run(zsrc[r], outs[r], Nz, zmode)
        return outs

    def partial(self, ins, d, direction):
        """execR2C/execC2R(out, in, d) for d = 1, 2 (enqueue_partial_*)"""
        Nx, Ny, Nz = self.shape
        P2, C, pls = self.P2, self.C, self.plans
        fwd = direction == dfft.FORWARD
        if fwd:
            outs = [np.full(n, np.nan + 0j, dtype=np.complex128) for n in self.nel]
        else:
            nin = [int(np.prod(pl.getInSize())) for pl in pls]
            outs = [np.full(n, np.nan, dtype=np.complex128 if self.c2c else np.float64) for n in nin]
        zmode = "c2c" if self.c2c else ("r2c" if fwd else "c2r")
        W = self.buffers(2)
        if d == 1:
            for r, pl in enumerate(pls):
                Pass(pl, "pz1" if fwd else "qz1").run(ins[r], outs[r], Nz, zmode)
            return outs
        first = [W[r][0] for r in range(self.P)]
        second = [W[r][1] if P2 > 1 else W[r][0] for r in range(self.P)]
        if fwd:
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "fz", c).run(ins[r], first[r], Nz, zmode)
                if P2 > 1:
                    self.exchange(dfft.FORWARD, 1, c, first, second)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "py2", c).run(second[r], outs[r], Ny)
        else:
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "qy2", c).run(ins[r], first[r], Ny)
                if P2 > 1:
                    self.exchange(dfft.INVERSE, 1, c, first, second)
            for c in range(C):
                for r, pl in enumerate(pls):
                    Pass(pl, "iz", c).run(second[r], outs[r], Nz, zmode)
        return outs
