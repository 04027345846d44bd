"""`make_comm(mode="auto")` tries the native RCCL transport in a child process first, so that a
collective that hangs costs a timeout instead of the job.  One GPU: the child forms a world of one
(ncclCommInitRank with nranks = 1), creates the transport and runs its round-trip self-test."""
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

from distributedfft_amd import torch_transport as tt  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_probe_child_runs_the_native_transport(monkeypatch):
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(free_port()))
    assert tt._probe_native(1, 1, timeout=240) is True


def test_probe_times_out_and_reports_failure():
    assert tt._probe_native(1, 1, timeout=2, cmd=[sys.executable, "-c", "import time; time.sleep(60)"]) is False
    assert tt._probe_native(1, 1, timeout=60, cmd=[sys.executable, "-c", "raise SystemExit(3)"]) is False
