#!/usr/bin/env python3
"""wall-clock per blocking exec at small sizes (launch-overhead check)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import distributedfft_amd as dfft
for N in (64, 128, 256, 512, 1024):
    plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), precision="double")
    plan.initFFT(dfft.GlobalSize(N, N, N), dfft.Partition(1, 1), True)
    x = torch.rand((N, N, N), dtype=torch.float64, device="cuda")
    out = torch.empty(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    for _ in range(5):
        plan.execR2C(out, x)
    t0 = time.perf_counter()
    it = 50
    for _ in range(it):
        plan.execR2C(out, x)
    dt = (time.perf_counter() - t0) / it * 1e3
    plan.enablePhaseTiming(True)
    plan.execR2C(out, x)
    ph = sum(v for _, v in plan.getPhaseTimes(dfft.FORWARD))
    print(f"R2C {N}^3 fp64: wall {dt:.4f} ms per exec, kernels {ph:.4f} ms   (reference V100-class: 64:n/a 128:0.168 256:1.158 512:10.41)")
