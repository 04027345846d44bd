"""Proof that the per-entry forward bound closes the round-5 tolerance hole (verdict item 2): the same fp64 single-rank transforms
on the A/B library (make -C distributedfft_amd/csrc exp; DFFT_AMD_LIBRARY) with the sound twiddle table and with the table rounded
through fp32 (DFFT_EXP_F32_TWIDDLES=1), both metrics side by side:
    old  max|got - want| / max|want|                    (bound 1e-11)
    new  max_k |got - want|[k] / max(|want[k]|, rms)    (tests/parity_metric.py; bound 1e-13 log2 n)
Run once per table setting (the switch is read when the library builds its first table):
    DFFT_AMD_LIBRARY=.../exp/libdfft_amd.so DFFT_EXP_F32_TWIDDLES=0|1 python tools/r6/twiddle_proof.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import distributedfft_amd as dfft  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from parity_metric import forward_bound, rms_rel  # noqa: E402

print(f"library {os.environ.get('DFFT_AMD_LIBRARY', 'default')}  DFFT_EXP_F32_TWIDDLES={os.environ.get('DFFT_EXP_F32_TWIDDLES', '0')}")
for N in (128, 256, 512, 1024):
    shape = (N, N, N)
    for center in (False, True):
        g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=2)
        if center:
            g -= 127.5 * (1 + 1j)
        plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
        plan.initFFT(dfft.GlobalSize(*shape), dfft.Pencil_Partition(1, 1), True, c2c=True)
        d_in = torch.from_numpy(g).cuda()
        d_out = torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        plan.execC2C(d_out, d_in, dfft.FORWARD)
        got = d_out[:g.size].cpu().numpy().reshape(shape)
        del d_in, d_out, plan
        orc.lib().orc_fft3d_c2c(g.ctypes.data_as(__import__("ctypes").c_void_p), *shape, -1)      # in place
        err = np.abs(got - g)
        del got
        mag = np.abs(g)
        old = float(err.max()) / float(mag.max())
        r = float(np.sqrt(np.vdot(g.reshape(-1), g.reshape(-1)).real / g.size))
        np.maximum(mag, r, out=mag)
        err /= mag
        new = float(err.max())
        del err, mag, g
        print(f"  {N}^3 {'zero-mean ' if center else 'uniform[0,255)'}: old {old:.3e} (bound 1e-11: {'PASS' if old < 1e-11 else 'FAIL'})   "
              f"per entry {new:.3e} (bound {forward_bound('double', N ** 3):.1e}: {'PASS' if new <= forward_bound('double', N ** 3) else 'FAIL'})", flush=True)
