"""Generates tests/golden/fft3d_small.npz: numpy.fft (pocketfft) transforms of the seeded
grids produced by the oracle's deterministic generator.  The reference itself cannot be run
here (needs nvcc/cuFFT), so these fixtures pin oracle + HIP path against an INDEPENDENT FFT.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as orc  # noqa: E402

SEED = 20260921
out = {"seed": np.int64(SEED)}
for shape in [(8, 8, 8), (12, 10, 14)]:
    tag = "x".join(map(str, shape))
    gc = orc.fill_block(shape, (0, 0, 0), shape, 2, SEED)
    gr = orc.fill_block(shape, (0, 0, 0), shape, 1, SEED)
    out["c2c_" + tag] = np.fft.fftn(gc)
    out["r2c_" + tag] = np.fft.rfftn(gr)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "fft3d_small.npz"), **out)
print("wrote", list(out))
