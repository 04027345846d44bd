"""native RCCL transport smoke test on one GPU: unique id, ncclCommInitRank(nranks=1), plan on it."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import distributedfft_amd as dfft
uid = dfft.Comm.rccl_unique_id()
print("unique id bytes", len(uid), uid[:8].hex())
comm = dfft.Comm.rccl(uid, 1, 0)
plan = dfft.MPIcuFFT_Pencil_Opt1(dfft.Configurations(), comm, precision="double")
plan.initFFT(dfft.GlobalSize(64, 64, 64), dfft.Pencil_Partition(1, 1), True, c2c=True)
x = torch.view_as_complex(torch.rand(64 ** 3, 2, dtype=torch.float64, device="cuda"))
out = torch.empty(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda")
back = torch.empty_like(x)
torch.cuda.synchronize()
plan.execC2C(out, x, dfft.FORWARD); plan.execC2C(back, out, dfft.INVERSE)
print("round trip", float((back / 64 ** 3 - x).abs().max()), "rank", plan.getRank(), "world", plan.getWorldSize())
del plan
comm.destroy()
print("rccl smoke ok")
