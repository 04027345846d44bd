"""More than one physical GPU: the native RCCL transport (grouped ncclSend/ncclRecv over xGMI inside libdfft_amd.so) and the
torch transport, one process per GPU, slab / pencil / Z_Then_YX against the oracle and round trips.  Skipped on boxes with a
single GPU (every gpurun box of rounds 1-2); written so that a multi-GPU lease runs it without changes:

    python -m pytest tests/test_gpu_multi_device.py -m gpu -q
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0

needs_two = pytest.mark.skipif(NDEV < 2, reason="needs at least two GPUs")


def launch(nproc, args, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_multi_device_worker.py")] + args
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)


def rccl_smoke(nproc):
    """tools/rccl_smoke.py: RCCL version, ncclCommCount, per-link and all-to-all rates through the library's transport -- first, so
    that a multi-GPU lease yields link numbers even if an FFT leg fails later"""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py"), "--gpus", str(nproc), "--mib", "64", "--iters", "3"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    print("RCCL_SMOKE", json.dumps(line))
    assert line["world_size"] == nproc and line["ncclCommCount"] == nproc and not line.get("errors")
    assert all(v > 0 for v in line["pair_shift_GBps"].values()) and line["alltoall_GBps_out_per_gpu"] > 0 and line["list_GBps_out_per_gpu"] > 0
    assert line["transport_counters"]["list"] >= 4
    return line


@needs_two
def test_0_rccl_smoke_on_all_gpus():
    line = rccl_smoke(min(8, NDEV))
    assert len(line["pair_shift_GBps"]) == min(8, NDEV) - 1


def test_rccl_smoke_on_one_gpu():
    """the same tool with a world of one (the own block through ncclSend / ncclRecv to itself, three duplicated communicators)"""
    line = rccl_smoke(1)
    assert line["duplicated_communicators"] == 3 and list(line["pair_shift_GBps"]) == ["0"]


@needs_two
@pytest.mark.parametrize("transport", ["rccl", "torch"])
@pytest.mark.parametrize("kind,grid", [("slab", "all"), ("pencil", "2xN"), ("zyx", "all")])
def test_one_process_per_gpu_against_the_oracle(kind, grid, transport):
    nproc = min(8, NDEV)
    if kind == "pencil" and nproc % 2:
        nproc -= 1
    out = launch(nproc, [kind, transport], 29700 + hash((kind, transport)) % 200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("MULTI_DEVICE_OK") == 1, out.stdout[-3000:]
    # the transport itself must report the number of ranks it spans (ncclCommCount for the native path)
    if transport == "rccl":
        assert f"rccl_nranks={nproc}" in out.stdout


@needs_two
def test_bench_runs_on_all_gpus():
    nproc = 2 if NDEV < 4 else (4 if NDEV < 8 else 8)
    # exactly the driver's command line: bench.py starts its own ranks
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--size", "256", "--steps", "3", "--warmup", "1"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    import json
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == nproc and line["round_trip_rel_linf"] < 1e-10
    assert line["config"]["alt"]["round_trip_rel_linf"] < 1e-10 if "alt" in line["config"] else True
    # the headline is the decomposition BASELINE.json names: pencil 2x2 / 2x4 at 4 / 8 GPUs (slab over all ranks is the alt)
    if nproc >= 4:
        assert line["config"]["decomposition"].startswith("pencil"), line["config"]["decomposition"]
        assert line["config"]["alt"]["decomposition"].startswith("slab")
    for name, m in line["xgmi"]["per_exchange_per_transform"].items():
        assert m["predicted_ms"] > 0 and m["links"] == m["group_ranks"] - 1
    assert "hidden_frac" in line["overlap"]


@pytest.mark.parametrize("kind,transport", [("slab", "rccl"), ("slab", "torch"), ("zyx", "rccl")])
def test_worker_runs_with_one_rank(kind, transport):
    """the worker script itself (process group, native RCCL communicator of size 1, oracle comparison) on a 1-GPU box; with
    one rank every class is the local transform, so this only pins the plumbing the multi-GPU cases above rely on (a 2 x N
    pencil grid needs two ranks: it is covered by the needs_two cases and, function-wise, by the virtual-rank tests)"""
    out = launch(1, [kind, transport], 29790 + ["slab", "zyx"].index(kind) + 2 * ["rccl", "torch"].index(transport))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "MULTI_DEVICE_OK" in out.stdout
    if transport == "rccl":
        assert "rccl_nranks=1" in out.stdout


def test_rccl_takes_virtual_memory_ranges_world_size_one():
    """buffers from the library's default backing (dfft_malloc(DFFT_CHUNK_DEFAULT): HIP virtual-memory ranges of several physical
    chunks) handed to RCCL itself: a communicator of one rank whose own block travels through ncclSend / ncclRecv to itself
    (transport option self_send) instead of a device copy.  This is what the exchange buffers of an N > 1 run are once the default
    backing is on everywhere; the real thing needs two GPUs (the needs_two cases above)."""
    import distributedfft_amd as dfft
    comm = dfft.Comm.rccl(dfft.Comm.rccl_unique_id(), 1, 0)
    assert comm.info() == (1, 1)
    comm.setOption("self_send", 1)
    n = 192 << 20                                   # 192 MiB: above the 32 MiB threshold of the default backing, three 64 MiB pieces
    a = dfft.DeviceBuffer.alloc(n, chunk_mib=64)
    b = dfft.DeviceBuffer.alloc(n)
    ta, tb = a.tensor(torch.int64), b.tensor(torch.int64)
    ta.copy_(torch.arange(n // 8, device="cuda", dtype=torch.int64))
    tb.zero_()
    torch.cuda.synchronize()
    # two pieces with a gap, as an exchange of a segmented buffer has them
    comm.alltoallv(0, a, [n // 2], [n // 4], b, [n // 2], [n // 8], [0], 0)
    torch.cuda.synchronize()
    lo, cnt = n // 8 // 8, n // 2 // 8
    assert torch.equal(tb[lo:lo + cnt], ta[n // 4 // 8:n // 4 // 8 + cnt])
    assert int(tb[:lo].abs().sum()) == 0 and int(tb[lo + cnt:].abs().sum()) == 0
    comm.destroy()
