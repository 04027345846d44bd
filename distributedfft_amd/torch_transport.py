"""Exchange transports for one-process-per-GPU runs under torch.distributed.

`make_comm` returns a dfft Comm for a P1 x P2 process grid (rank = i*P2 + j, the reference's
pidx = pidx_i*P2 + pidx_j, src/pencil/mpicufft_pencil_opt1.cpp:67-68):

  * "rccl"  -- the library's own RCCL communicator (grouped ncclSend/ncclRecv over xGMI inside
               libdfft_amd.so); torch.distributed only broadcasts the ncclUniqueId.
  * "torch" -- torch.distributed.all_to_all_single on row/column process groups (backend
               "nccl" = RCCL on ROCm, "gloo" on CPU), driven through the C ABI's callback
               transport.  Buffers the library exchanges must be registered so raw pointers
               can be mapped back to tensors.
  * "auto"  -- rccl if it can be created and passes a small round-trip self-test on every
               rank, otherwise torch.  With more than one rank the native transport is first
               tried in a short-lived child process per rank (`_probe_native`), so that a hang
               inside a collective -- which no exception handler can catch -- costs a timeout
               instead of the job.

Stream contract: the plan must run on torch's *current* stream while the callback executes, i.e.
create `side = torch.cuda.Stream()`, `plan.setStream(side.cuda_stream)` and call exec inside
`with torch.cuda.stream(side):`.  (Launching on the legacy null stream next to torch's default
stream was observed to race with gloo's staging copies on ROCm; a dedicated stream is exact.)
"""
import os
import subprocess
import sys

import torch

from . import api


class TorchComm:
    def __init__(self, dist, rank, world, P1, P2, list_callback=True):
        """list_callback = False: no point-to-point schedule callback -- the library then runs a relay hop as group - 1 all-to-all-v
        layers through `_alltoallv` (what a transport without a native schedule does; tests)"""
        self.dist, self.rank, self.world = dist, rank, world
        self.groups = {}
        # every rank creates every group, in the same order (torch.distributed requirement)
        for i in range(P1):
            ranks = [i * P2 + j for j in range(P2)]
            self.groups[tuple(ranks)] = dist.new_group(ranks) if len(ranks) > 1 else None
        for j in range(P2):
            ranks = [i * P2 + j for i in range(P1)]
            self.groups[tuple(ranks)] = dist.new_group(ranks) if len(ranks) > 1 else None
        # the whole world: the two-hop relay (option "relay") runs world-wide all-to-alls; None = the default group
        self.groups.setdefault(tuple(range(world)), None)
        self.buffers = []
        self.comm = api.Comm.callback(world, rank, self._alltoallv)
        if list_callback:
            self.comm.setListCallback(self._sendrecv_list)
        self.calls = 0
        self.p2p_calls = 0
        self.list_calls = 0

    def _device(self):
        return self.buffers[0].device if self.buffers else torch.device("cpu")

    def register(self, tensor):
        """make a tensor's storage known to the pointer -> tensor lookup"""
        self.buffers.append(tensor.reshape(-1).view(torch.uint8))

    def _slice(self, ptr, nbytes):
        for t in self.buffers:
            base = t.data_ptr()
            if base <= ptr and ptr + nbytes <= base + t.numel():
                return t[ptr - base: ptr - base + nbytes]
        # memory the library owns (the relay's staging and table buffers): wrap the raw pointer
        if self.buffers and self.buffers[0].is_cuda:
            return torch.as_tensor(api.DeviceBuffer(ptr, nbytes, owned=False), device="cuda")
        import ctypes
        import numpy as np
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))

    def _p2p(self, send, sc, sd, recv, rc, rd, group, me, pg):
        """peer blocks that are not laid out back to back (the relay's pieces): one send / receive per peer"""
        ops, landed = [], []
        # gloo moves device tensors in its collectives but not in point-to-point operations: stage those through the host
        stage = bool(self.buffers and self.buffers[0].is_cuda) and self.dist.get_backend(pg) == "gloo"
        for q, peer in enumerate(group):
            if q == me:
                if rc[q]:
                    self._slice(recv + rd[q], rc[q]).copy_(self._slice(send + sd[q], sc[q]))
                continue
            if sc[q]:
                t = self._slice(send + sd[q], sc[q])
                ops.append(self.dist.P2POp(self.dist.isend, t.cpu() if stage else t, peer, group=pg))
            if rc[q]:
                t = self._slice(recv + rd[q], rc[q])
                if stage:
                    h = torch.empty(rc[q], dtype=torch.uint8)
                    landed.append((t, h))
                    t = h
                ops.append(self.dist.P2POp(self.dist.irecv, t, peer, group=pg))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        for t, h in landed:
            t.copy_(h)
        self.p2p_calls += 1

    def _sendrecv_list(self, sends, recvs, stream):
        """a point-to-point schedule over the whole world in ONE batch (dfft_sendrecv_list_fn: a hop of the two-hop relay -- several
        pieces per peer, matched in list order at both ends of a link)"""
        pg = self.groups[tuple(range(self.world))]
        on_device = bool(self.buffers and self.buffers[0].is_cuda)
        # gloo moves device tensors in its collectives but not in point-to-point operations: stage those through the host
        stage = on_device and self.dist.get_backend(pg) == "gloo"

        def run():
            ops, landed = [], []
            for peer, ptr, nb in sends:
                t = self._slice(ptr, nb)
                ops.append(self.dist.P2POp(self.dist.isend, t.cpu() if stage else t, peer, group=pg))
            for peer, ptr, nb in recvs:
                t = self._slice(ptr, nb)
                if stage:
                    h = torch.empty(nb, dtype=torch.uint8)
                    landed.append((t, h))
                    t = h
                ops.append(self.dist.P2POp(self.dist.irecv, t, peer, group=pg))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()
            for t, h in landed:
                t.copy_(h)

        if on_device:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0)):
                run()
        else:
            run()
        self.list_calls += 1

    def _alltoallv(self, send, sc, sd, recv, rc, rd, group, me, stream):
        # Which collective: decided by WHAT the library is doing, which is the same on every rank of the group -- never by this
        # rank's tables alone (round-5 advice: ranks that choose differently hang).  A plan's exchange has its peer blocks back to
        # back in rank order on every rank by construction: one all_to_all_single.  The layers of a relay schedule run through the
        # default sendrecv_list (counter "layered" > 0 while it does) are pieces in unrelated allocations: one send / receive per peer.
        layered = self.comm.getCounter("layered") > 0
        packed = not layered
        if packed:
            nzs, nzr = [q for q in range(len(group)) if sc[q]], [q for q in range(len(group)) if rc[q]]
            ok = all(sd[b] == sd[a] + sc[a] for a, b in zip(nzs, nzs[1:])) and all(rd[b] == rd[a] + rc[a] for a, b in zip(nzr, nzr[1:]))
            if not (ok and all(c % 8 == 0 for c in sc + rc)):
                raise RuntimeError("exchange table with peer blocks that are not back to back in rank order (or not multiples of 8 bytes): "
                                   f"counts {sc} / {rc}, displacements {sd} / {rd}")
            first_s, first_r = (nzs or [0])[0], (nzr or [0])[0]
            # (a rank with nothing to send or receive still takes part, with an empty tensor)
            s = self._slice(send + sd[first_s], sum(sc)).view(torch.int64) if sum(sc) else torch.empty(0, dtype=torch.int64, device=self._device())
            r = self._slice(recv + rd[first_r], sum(rc)).view(torch.int64) if sum(rc) else torch.empty(0, dtype=torch.int64, device=self._device())
        on_device = bool(self.buffers and self.buffers[0].is_cuda)

        def run():
            if packed:
                self.dist.all_to_all_single(r, s, output_split_sizes=[c // 8 for c in rc],
                                            input_split_sizes=[c // 8 for c in sc], group=self.groups[tuple(group)])
            else:
                self._p2p(send, sc, sd, recv, rc, rd, group, me, self.groups[tuple(group)])

        if on_device:
            # the library hands over the HIP stream this exchange is ordered on (its communication
            # stream); torch collectives order themselves against torch's *current* stream
            with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0)):
                run()
        else:
            run()
        self.calls += 1

    # duck-typing so a TorchComm can be passed wherever a Comm is expected
    def setOption(self, key, value):
        """transport knobs of the communicator underneath (e.g. 'relay', include/dfft_c.h: dfft_comm_set_option)"""
        self.comm.setOption(key, value)

    def info(self):
        return self.comm.info()

    @property
    def _h(self):
        return self.comm._h

    @property
    def nranks(self):
        return self.world


def _selftest(dist, comm, rank, world, P1, P2):
    """tiny forward+inverse through the transport; returns max round-trip error over ranks"""
    n = 32
    plan = api.MPIcuFFT_Pencil_Opt1(api.Configurations(), comm, precision="double", rank=rank)
    plan.initFFT(api.GlobalSize(n, n, n), api.Partition(P1, P2), allocate=False, c2c=True)
    side = torch.cuda.Stream()
    plan.setStream(side.cuda_stream)
    work = torch.empty(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
    plan.setWorkArea(work)
    s = plan.getInSize()
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    x = torch.view_as_complex(torch.rand((s[0] * s[1] * s[2], 2), dtype=torch.float64, device="cuda", generator=g))
    out = torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    back = torch.zeros_like(x)
    if isinstance(comm, TorchComm):
        comm.register(work)
        comm.register(out)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        plan.execC2C(out, x, api.FORWARD)
        plan.execC2C(back, out, api.INVERSE)
    torch.cuda.synchronize()
    err = ((back / float(n) ** 3 - x).abs().max() / x.abs().max()).reshape(1)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    return float(err.item())


_PROBED = {}


def _probe_native(P1, P2, timeout=None, cmd=None):
    """Create the native RCCL transport and run its self-test in a child process that joins the
    other ranks' children on its own rendezvous port.  Returns True only if the child exits
    cleanly within `timeout` seconds; a child that hangs is killed (by pid)."""
    timeout = float(os.environ.get("DFFT_PROBE_TIMEOUT", "150")) if timeout is None else timeout
    env = dict(os.environ)
    port = int(env.get("MASTER_PORT", "29500"))
    env["MASTER_PORT"] = str(port + 23 if port + 23 < 65536 else port - 23)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    if cmd is None:
        cmd = [sys.executable, "-m", "distributedfft_amd.torch_transport", "--probe", str(P1), str(P2)]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        _, err = child.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        print(f"[rank {env.get('RANK', '0')}] native RCCL probe did not finish in {timeout:.0f} s", flush=True)
        return False
    if child.returncode != 0:
        tail = err.decode(errors="replace").strip().splitlines()[-1:] if err else []
        print(f"[rank {env.get('RANK', '0')}] native RCCL probe failed (rc {child.returncode}) {' '.join(tail)}", flush=True)
    return child.returncode == 0


def _probe_main(argv):
    """child side of _probe_native: its own process group, the native transport, the self-test"""
    import torch.distributed as dist
    P1, P2 = int(argv[0]), int(argv[1])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    dist.init_process_group("nccl", rank=rank, world_size=world)
    make_comm(dist, rank, world, P1, P2, mode="rccl", probe=False)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def make_comm(dist, rank, world, P1, P2, mode="auto", probe=None):
    if mode == "auto" and (probe if probe is not None else (world > 1 and os.environ.get("DFFT_TRANSPORT_PROBE", "1") != "0")):
        if world not in _PROBED:      # once per process: later plans (other grids) reuse the verdict
            fine = torch.tensor([1.0 if _probe_native(P1, P2) else 0.0], device="cuda")
            dist.all_reduce(fine, op=dist.ReduceOp.MIN)
            _PROBED[world] = fine.item() > 0
        if not _PROBED[world]:
            mode = "torch"
    if mode in ("auto", "rccl"):
        ok = torch.ones(1, device="cuda")
        comm = None
        try:
            ids = [api.Comm.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = api.Comm.rccl(ids[0], world, rank)
        except Exception as e:   # noqa: BLE001
            if mode == "rccl":
                raise
            print(f"[rank {rank}] native RCCL transport unavailable ({e}); using torch transport", flush=True)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() > 0:
            try:
                err = _selftest(dist, comm, rank, world, P1, P2)
            except Exception as e:   # noqa: BLE001  (a rank-local failure; the others learn it below)
                if mode == "rccl":
                    raise
                print(f"[rank {rank}] native RCCL self-test raised {e!r}", flush=True)
                err = float("inf")
            good = torch.tensor([1.0 if err < 1e-10 else 0.0], device="cuda")
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if good.item() > 0:
                return comm, "rccl (native grouped send/recv)"
            if mode == "rccl":
                raise RuntimeError(f"native RCCL transport failed its self-test: round trip {err}")
    tc = TorchComm(dist, rank, world, P1, P2)
    return tc, "torch"


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--probe":
        sys.exit(_probe_main(sys.argv[2:]))
    sys.exit("usage: python -m distributedfft_amd.torch_transport --probe P1 P2")
