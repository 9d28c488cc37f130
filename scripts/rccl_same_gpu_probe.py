import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def run(rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29811", RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
    t = torch.ones(4, device="cuda:0") * (rank + 1)
    dist.all_reduce(t)
    print("rank", rank, t.tolist(), flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(run, nprocs=2, join=True)
