#!/usr/bin/env python3
"""Context numbers, NOT part of the product path: the vendor FFT (rocFFT through torch.fft) and this
library on the same grid, same GPU.  usage: python tools/vendor_compare.py [N] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import distributedfft_amd as dfft  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = N ** 3


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


x = torch.view_as_complex(torch.rand((n, 2), dtype=torch.float64, device="cuda")).reshape(N, N, N)
res = {}
try:
    res["rocFFT (torch.fft.fftn) c2c forward ms"] = timeit(lambda: torch.fft.fftn(x), iters)
except Exception as e:  # noqa: BLE001
    res["rocFFT c2c"] = f"failed: {e}"
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), True, c2c=True)
out = torch.empty(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
res["this library c2c forward ms"] = timeit(lambda: plan.execC2C(out, x, dfft.FORWARD), iters)
back = torch.empty_like(x)
res["this library c2c inverse ms"] = timeit(lambda: plan.execC2C(back, out, dfft.INVERSE), iters)
del plan, out, back
xr = torch.rand((N, N, N), dtype=torch.float64, device="cuda")
try:
    res["rocFFT (torch.fft.rfftn) r2c forward ms"] = timeit(lambda: torch.fft.rfftn(xr), iters)
except Exception as e:  # noqa: BLE001
    res["rocFFT r2c"] = f"failed: {e}"
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), True)
out = torch.empty(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
plan.enablePhaseTiming(True)
res["this library r2c forward ms (execR2C)"] = timeit(lambda: plan.execR2C(out, xr), iters)
res["  r2c phases"] = plan.getPhaseTimes(dfft.FORWARD)
yr = torch.empty_like(xr)
res["this library c2r inverse ms (execC2R)"] = timeit(lambda: (plan.execR2C(out, xr), plan.execC2R(yr, out)), iters)
res["  c2r phases"] = plan.getPhaseTimes(dfft.INVERSE)
for k, v in res.items():
    print(f"{k:46s} {v}")
