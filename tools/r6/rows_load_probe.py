"""What would an inverse y pass cost that reads natural rows [x][ky][kz'] (129-wide, 2064-byte pitch) instead of the tiled
blocks [x][kz'/TL][ky][kz'%TL]?  (Round-5 verdict item 4a: an x^-1 that reads the API layout as flat aligned runs leaves exactly that
layout behind.)  The partial inverse transform d = 2 of the pencil classes already has such a pass (build_pipeline: qy2, LOAD_KMAJOR
with KS = zs, same transposed-tile store as the full inverse's y^-1), so its span next to the full inverse's y^-1 span is the answer.
Rank 0 of 2 x 4, 1024^3 fp64 R2C, exchange stubbed, one compute stream."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import distributedfft_amd as dfft  # noqa: E402

N = 1024
for chunks in (4, 1):
    stub = dfft.Comm.callback(8, 0, lambda *a: None)
    pl = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), stub, precision="double", rank=0)
    pl.setOption("compute_streams", 1)
    pl.setOption("pipeline_chunks", chunks)
    pl.initFFT(dfft.GlobalSize(N, N, N), dfft.Pencil_Partition(2, 4), True)
    isz = pl.getInSize()
    x = torch.rand(isz, dtype=torch.float64, device="cuda")
    out = torch.zeros(pl.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
    back = torch.zeros_like(x)
    torch.cuda.synchronize()
    pl.enablePhaseTiming(True)
    acc = {}
    for it in range(12):
        pl.execR2C(out, x)
        pl.execC2R(back, out)
        full = dict(pl.getPhaseTimes(dfft.INVERSE))
        pl.execC2R(back, out, 2)
        part = dict(pl.getPhaseTimes(dfft.INVERSE))
        if it >= 2:
            for k, v in full.items():
                acc[("full", k)] = acc.get(("full", k), 0.0) + v / 10
            for k, v in part.items():
                acc[("d=2", k)] = acc.get(("d=2", k), 0.0) + v / 10
    import time
    pl.enablePhaseTiming(False)
    walls = {}
    for name, fn in (("full inverse", lambda: pl.execC2R(back, out)), ("d = 2 inverse (rows-load y^-1 + z^-1)", lambda: pl.execC2R(back, out, 2))):
        pl.execR2C(out, x)
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        walls[name] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"chunks = {pl.getPipelineChunks()}")
    for k, v in walls.items():
        print(f"  wall {k}: {v:.3f} ms per blocking call")
    for k, v in acc.items():
        if v > 0:
            print(f"  {k[0]:5s} {k[1]:12s} {v:7.3f} ms")
    del pl
    stub.destroy()
