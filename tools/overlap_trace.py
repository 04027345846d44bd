#!/usr/bin/env python3
"""Workload for the overlap evidence in profiles/: a pencil 2x2 transform on four virtual ranks
(threads) sharing the GPU, pipelined (DFFT_CHUNKS, default 4).  Run under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o ov -- python tools/overlap_trace.py
and summarise with tools/overlap_summary.py DIR."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import distributedfft_amd as dfft
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P1, P2 = 2, 2
world = dfft.Comm.local(P1 * P2)
ranks = []
for r in range(P1 * P2):
    pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), world, precision="double", rank=r)
    pl.initFFT(dfft.GlobalSize(N, N, N), dfft.Pencil_Partition(P1, P2), True, c2c=True)
    s = pl.getInSize()
    x = torch.view_as_complex(torch.rand((s[0] * s[1] * s[2], 2), dtype=torch.float64, device="cuda"))
    out = torch.empty(pl.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    ranks.append((pl, x, out, torch.empty_like(x)))
torch.cuda.synchronize()
print("pipeline chunks", ranks[0][0].getPipelineChunks())
with ThreadPoolExecutor(len(ranks)) as ex:
    for it in range(3):
        list(ex.map(lambda t: t[0].execC2C(t[2], t[1], dfft.FORWARD), ranks))
        list(ex.map(lambda t: t[0].execC2C(t[3], t[2], dfft.INVERSE), ranks))
torch.cuda.synchronize()
err = max(float((t[3] / float(N) ** 3 - t[1]).abs().max()) for t in ranks)
print("round trip", err)
