#!/usr/bin/env python3
"""Per-phase device times (HIP events on the plan's stream) of a single-GPU 3-D C2C transform.
usage: python tools/phase_times.py [N] [precision] [iters]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import distributedfft_amd as dfft  # noqa: E402

_a = sys.argv[1] if len(sys.argv) > 1 else "1024"
Nx, Ny, Nz = ([int(v) for v in _a.split("x")] if "x" in _a else [int(_a)] * 3)
N = Nx
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
esz = 16 if prec == "double" else 8
cdt = torch.complex128 if prec == "double" else torch.complex64
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision=prec)
plan.initFFT(dfft.GlobalSize(Nx, Ny, Nz), dfft.Partition(1, 1), True, c2c=True)
n = Nx * Ny * Nz
x = torch.view_as_complex(torch.rand((n, 2), dtype=torch.float64 if prec == "double" else torch.float32,
                                     device="cuda") * 255)
out = torch.empty(plan.getDomainSize() // esz, dtype=cdt, device="cuda")
back = torch.empty_like(x)
torch.cuda.synchronize()
plan.enablePhaseTiming(True)
acc = {}
for it in range(iters + 1):
    plan.execC2C(out, x, dfft.FORWARD)
    f = plan.getPhaseTimes(dfft.FORWARD)
    plan.execC2C(back, out, dfft.INVERSE)
    b = plan.getPhaseTimes(dfft.INVERSE)
    if it == 0:
        continue
    for k, v in f + b:
        acc.setdefault(k, []).append(v)
vol = 2.0 * esz * n
print(f"N={Nx}x{Ny}x{Nz} {prec} variant={os.environ.get('DFFT_VARIANT', '0')}")
tot = 0
for k, v in acc.items():
    if "FFT" in k:
        m = sum(v) / len(v)
        tot += m
        print(f"  {k:10s} {m:8.3f} ms  min {min(v):8.3f}  {vol / m / 1e6:8.1f} GB/s")
print(f"  total fft {tot:.3f} ms -> {2 * 5 * n * __import__('math').log2(n) / tot / 1e6:.0f} GFLOP/s")
