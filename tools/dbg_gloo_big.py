import os, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
for n in (1 << 17, 1 << 20, 1 << 23):
    s = (torch.arange(2 * n, dtype=torch.int64, device="cuda") + (rank << 40))
    r = torch.full((2 * n,), -1, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(r, s, output_split_sizes=[n, n], input_split_sizes=[n, n])
    torch.cuda.synchronize()
    exp = torch.cat([torch.arange(rank * n, rank * n + n, dtype=torch.int64, device="cuda") + (q << 40) for q in range(2)])
    print(rank, n * 8 >> 20, "MiB per split ok:", bool((r == exp).all()), "bad:", int((r != exp).sum()), flush=True)
dist.barrier(); dist.destroy_process_group()
