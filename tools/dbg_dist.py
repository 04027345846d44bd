import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import test_gpu_parity as T
from oracle import oracle as orc
shape = tuple(int(v) for v in sys.argv[1].split("x")); P1, P2 = int(sys.argv[2]), int(sys.argv[3]); real = sys.argv[4] == "r"
if real:
    plans, ins, spec, backs = T.run_distributed_real(shape, P1, P2, "double")
    g = orc.fill_block(shape, (0, 0, 0), shape, 1, seed=13); want = np.fft.rfftn(g)
else:
    plans, ins, spec, backs = T.run_distributed(shape, P1, P2, "double")
    g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=7); want = np.fft.fftn(g)
for r, pl in enumerate(plans):
    s, o = pl.getOutSize(), pl.getOutStart()
    ref = want[:, o[1]:o[1] + s[1], o[2]:o[2] + s[2]]
    d = np.abs(spec[r] - ref) / np.max(np.abs(want))
    bad = np.argwhere(d > 1e-10)
    print("rank", r, "fwd", d.max(), "nbad", len(bad), "first bad", bad[:3].tolist(),
          "rt", np.max(np.abs(backs[r] / np.prod(shape) - ins[r])) / 255)
