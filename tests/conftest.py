import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
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
