"""N > 1 host path on CPU: world_size 2 (slab) and 4 (pencil 2x2, uneven) over gloo."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("P1,P2,shape,seq", [(2, 1, "16x8x8", ""), (1, 2, "8x8x16", ""), (2, 2, "16x16x8", ""), (2, 2, "8x8x16", ""),
                                             (3, 2, "16x16x16", ""), (3, 1, "16x16x8", ""),
                                             (2, 1, "16x8x8", "zyx"), (3, 1, "16x6x10", "zyx"), (2, 1, "8x8x16", "yzx"), (3, 1, "10x16x6", "yzx")])
def test_gloo_exchange_path(P1, P2, shape, seq):
    run_world(P1, P2, shape, seq)


@pytest.mark.parametrize("relay", [1, 2, 3])
@pytest.mark.parametrize("P1,P2,shape", [(2, 2, "16x16x8"), (2, 3, "12x18x16"), (3, 2, "18x16x10"), (2, 2, "10x6x14")])
def test_gloo_relayed_exchange_path(P1, P2, shape, relay):
    """the two-hop relay (dfft_comm_set_option "relay": bit 0 = exchange 2, bit 1 = exchange 1) on the torch transport over gloo:
    every group exchange becomes world-wide all-to-alls of message parts; the blocks must arrive exactly where the direct
    exchange puts them (the worker checks the transform, the mirror property of the inverse exchange and the call counts).
    Uneven splits (10 = 5 + 5 rows of 6 / 14 columns, 18 = 6 + 6 + 6 ...), groups of 2 and 3, worlds of 4 and 6."""
    run_world(P1, P2, shape, "", {"DFFT_TEST_RELAY": str(relay)})


@pytest.mark.parametrize("P1,P2,shape,relay", [(2, 2, "16x16x8", 3), (2, 3, "12x18x16", 1), (3, 2, "18x16x10", 2)])
def test_gloo_relayed_exchange_on_a_transport_without_schedules(P1, P2, shape, relay):
    """a transport that has no point-to-point schedule of its own (here: the torch transport with its list callback left out; in the
    product: the host-staged MPI shim) runs every hop of the relay as group - 1 all-to-all-v layers (dfft_comm::sendrecv_list's
    default): same bytes in the same places, counted by dfft_comm_get_counter"""
    run_world(P1, P2, shape, "", {"DFFT_TEST_RELAY": str(relay), "DFFT_TEST_NO_LIST": "1"})


def run_world(P1, P2, shape, seq, extra_env=None):
    world = P1 * P2
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(P1), str(P2), shape] + ([seq] if seq else []),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r} ok" in out


def _clean_env():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` as the driver calls it -- NOT inside torch.distributed.run -- starts its N ranks itself
    (the reference's launcher builds its own `mpiexec -n P` line, launch.py:168-247).  Without a GPU the ranks can only meet:
    --rendezvous-only has them agree on the world over gloo and leave."""
    import json
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--rendezvous-only"], capture_output=True, text=True,
                         timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"rendezvous": "ok", "world_size": 3, "sum_of_ranks_plus_1": 6, "self_launched": True}


def test_bench_fails_loudly_when_a_rank_fails():
    """a rank that cannot run (here: no GPU) makes the self-launched job return non-zero and print no result line"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU: the ranks would run")
    root = os.path.dirname(HERE)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "64", "--steps", "1",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
