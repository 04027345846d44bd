import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
# 1. raw all_to_all_single on cuda tensors with uneven splits
s = torch.arange(10, dtype=torch.int64, device="cuda") + 100 * rank
r = torch.full((16,), -1, dtype=torch.int64, device="cuda")
splits_in = [3, 7] if rank == 0 else [4, 6]
splits_out = [3, 4] if rank == 0 else [7, 6]
r = r[:sum(splits_out)]
dist.all_to_all_single(r, s, output_split_sizes=splits_out, input_split_sizes=splits_in)
torch.cuda.synchronize()
print(rank, "raw:", r.tolist(), flush=True)
# 2. view-of-buffer semantics
buf = torch.zeros(256, dtype=torch.uint8, device="cuda")
rv = buf[16:16 + 8 * sum(splits_out)].view(torch.int64)
dist.all_to_all_single(rv, s, output_split_sizes=splits_out, input_split_sizes=splits_in)
torch.cuda.synchronize()
print(rank, "view:", buf[16:16 + 8 * sum(splits_out)].view(torch.int64).tolist(), flush=True)
# 3. full plan vs oracle
import numpy as np
import distributedfft_amd as dfft
from distributedfft_amd.torch_transport import TorchComm
from oracle import oracle as orc
shape = tuple([int(os.environ.get("DBG_N", "16"))] * 3)
tc = TorchComm(dist, rank, world, world, 1)
if os.environ.get("DBG_SYNC"):
    _orig = tc._alltoallv
    def _synced(*a):
        torch.cuda.synchronize(); _orig(*a); torch.cuda.synchronize()
    tc._alltoallv = _synced
    tc.comm = dfft.Comm.callback(world, rank, tc._alltoallv)
plan = dfft.MPIcuFFT_Slab_Opt1(dfft.Configurations(), tc, precision="double", rank=rank)
plan.initFFT(dfft.GlobalSize(*shape), dfft.Partition(world, 1), allocate=False, c2c=True)
side = torch.cuda.Stream(); plan.setStream(side.cuda_stream)
work = torch.zeros(plan.getWorkSizeDevice(), dtype=torch.uint8, device="cuda")
plan.setWorkArea(work); tc.register(work)
isz, ist = plan.getInSize(), plan.getInStart()
blk = orc.fill_block(shape, ist, isz, 2, seed=3)
d_in = torch.from_numpy(blk).cuda()
d_out = torch.zeros(plan.getDomainSize() // 16, dtype=torch.complex128, device="cuda"); tc.register(d_out)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    plan.execC2C(d_out, d_in, dfft.FORWARD)
g = orc.fill_block(shape, (0, 0, 0), shape, 2, seed=3)
want = np.fft.fftn(g)
s_, o_ = plan.getOutSize(), plan.getOutStart()
got = d_out[:s_[0] * s_[1] * s_[2]].cpu().numpy().reshape(s_)
ref = want[:, o_[1]:o_[1] + s_[1], o_[2]:o_[2] + s_[2]]
print(rank, "fwd err", np.max(np.abs(got - ref)) / np.max(np.abs(want)), "calls", tc.calls, flush=True)
d_back = torch.zeros_like(d_in)
for it in range(3):
    with torch.cuda.stream(side):
        plan.execC2C(d_out, d_in, dfft.FORWARD)
        plan.execC2C(d_back, d_out, dfft.INVERSE)
    rt = (d_back / float(shape[0]) ** 3 - d_in).abs().max() / d_in.abs().max()
    print(rank, "iter", it, "round trip", float(rt), "calls", tc.calls, flush=True)
dist.barrier(); dist.destroy_process_group()
