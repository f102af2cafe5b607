import os, sys, torch, torch.distributed as dist, datetime
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
    t = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "allreduce ->", t.tolist(), flush=True)
    if rank == 0:
        dist.send(torch.arange(4.0, device="cuda"), 1)
    else:
        r = torch.empty(4, device="cuda"); dist.recv(r, 0); torch.cuda.synchronize(); print("recv", r.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:600], flush=True)
