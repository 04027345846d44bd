#!/usr/bin/env python3
"""Per-phase device times of a single-GPU execR2C / execC2R (HIP events on the plan's stream).
usage: python tools/phase_times_r2c.py [N] [precision] [iters]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import distributedfft_amd as dfft  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
esz = 16 if prec == "double" else 8
rdt, cdt = (torch.float64, torch.complex128) if prec == "double" else (torch.float32, torch.complex64)
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision=prec)
plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), True)
x = torch.rand(N, N, N, dtype=rdt, device="cuda") * 255
out = torch.empty(plan.getDomainSize() // esz, dtype=cdt, device="cuda")
back = torch.empty_like(x)
torch.cuda.synchronize()
plan.enablePhaseTiming(True)
acc = {}
for it in range(iters + 1):
    plan.execR2C(out, x)
    f = plan.getPhaseTimes(dfft.FORWARD)
    plan.execC2R(back, out)
    b = plan.getPhaseTimes(dfft.INVERSE)
    if it:
        for k, v in f + b:
            acc.setdefault(k, []).append(v)
torch.cuda.synchronize()
print(f"round trip rel L-inf {float((back / float(N) ** 3 - x).abs().max()) / 255.0:.2e}")
Nzc = N // 2 + 1
real_b, half_b = N ** 3 * esz / 2, N * N * Nzc * esz
print(f"N={N} {prec} R2C/C2R  (real {real_b / 2**30:.2f} GiB, half spectrum {half_b / 2**30:.2f} GiB)")
for k, v in acc.items():
    if "FFT" in k:
        m = sum(v) / len(v)
        vol = real_b + half_b if k.startswith("z") else 2 * half_b
        print(f"  {k:10s} {m:8.3f} ms  min {min(v):8.3f}  {vol / m / 1e6:8.1f} GB/s")
