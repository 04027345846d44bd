import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "baseline_config: parity test of a BASELINE.json configuration at its real size; on an MI355X "
                            "(>= 250 GiB of HBM) a skip of such a test is reported as a FAILURE")
    config.addinivalue_line("markers", "slow: the part of a long parametrisation that the default GPU run leaves out (see THINNED below); "
                            "run everything with -m \"gpu and slow\" next to -m gpu, or DFFT_TEST_SLOW=1")
    # Built artefacts are not in git history.  On a fresh checkout compile them once (hipcc
    # cross-compiles gfx950 without a GPU; gcc for the oracle) -- the same thing
    # __graft_entry__.build() does.  The package itself never builds or falls back at import time.
    lib = os.path.join(ROOT, "distributedfft_amd", "libdfft_amd.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C",
                               os.path.join(ROOT, "distributedfft_amd", "csrc")])
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])


# The GPU suite has to stay well inside the driver's step limit (round 5: 566 s of 1200 s).  ONE function sweeps a long cross product
# of a single code path -- the forced z, x, y single-rank order: 16 shapes x 5 private-layout settings x 2 precisions = 160 instances --
# and only that one is thinned, by an explicit rule on its parameters (round-5 advice: not by collection order): the default layout
# setting (1, 128) runs every shape in both precisions; each other setting runs one shape per kernel family (ragged power of two,
# mixed radix, Bluestein, sub-tile workgroups, a 2048-point axis).  The rest carries the marker `slow`:
#     python -m pytest tests -m gpu            (what the driver runs)        python -m pytest tests -m "gpu and slow"   (the rest)
# Everything else that round 5 thinned (relay bit-identity, two-level and long-Bluestein sweeps) is back in the default run.
ZXY_FAMILY_SHAPES = {(32, 64, 24), (12, 20, 24), (17, 33, 9), (4096, 4, 16), (16, 2048, 24)}


def _zxy_is_slow(params):
    return (params["layout"], params["pad"]) != (1, 128) and tuple(params["shape"]) not in ZXY_FAMILY_SHAPES


THINNED = {"test_single_order_zxy_forced_vs_oracle": _zxy_is_slow}


def pytest_collection_modifyitems(config, items):
    for it in items:
        name = getattr(it, "originalname", None) or it.name.split("[")[0]
        rule = THINNED.get(name)
        if rule is not None and hasattr(it, "callspec") and rule(it.callspec.params):
            it.add_marker(pytest.mark.slow)
    expr = config.getoption("-m") or ""
    if "slow" in expr or os.environ.get("DFFT_TEST_SLOW") == "1":
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("slow") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def _hbm_total_gib():
    try:
        import torch
        if not torch.cuda.is_available():
            return 0.0
        return torch.cuda.mem_get_info()[1] / 2 ** 30
    except Exception:  # noqa: BLE001
        return 0.0


@pytest.fixture(autouse=True)
def _release_device_memory_around_large_tests(request):
    """BASELINE-sized tests need most of the 288 GB: drop what earlier tests left in torch's caching allocator (and the plans
    whose work areas the library owns) before their memory gates look at the free figure, and again afterwards."""
    big = request.node.get_closest_marker("baseline_config") is not None
    if big:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    yield
    if big:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """A test of a BASELINE configuration that skips on a box able to run it is a failure, not a skip (round 3: the 2048^3
    test skipped by its own memory gate in every log while the documents called it green)."""
    outcome = yield
    rep = outcome.get_result()
    if rep.skipped and not hasattr(rep, "wasxfail") and item.get_closest_marker("baseline_config") is not None and _hbm_total_gib() >= 250:
        why = rep.longrepr[2] if isinstance(rep.longrepr, tuple) else str(rep.longrepr)
        rep.outcome = "failed"
        rep.longrepr = f"BASELINE configuration test skipped on a box with {_hbm_total_gib():.0f} GiB of HBM: {why}"
