"""Probe (round 6, session 7): do gloo's collectives take CUDA tensors when two ranks share ONE GPU?  (all_gather_into_tensor async, all_reduce, barrier,
all_gather of a list: yes, through the host) -- what bench.py's test knobs NGF_BENCH_BACKEND=gloo NGF_BENCH_ONE_DEVICE=1 rest on.   python profiles/probe_gloo_cuda.py"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    send = torch.full((1000,), float(rank + 1), device=dev)
    recv = torch.empty((world * 1000,), device=dev)
    try:
        wk = dist.all_gather_into_tensor(recv, send, async_op=True); wk.wait(); torch.cuda.synchronize()
        print(rank, "all_gather_into_tensor cuda ok", recv[::1000].tolist(), flush=True)
    except Exception as e:
        print(rank, "all_gather_into_tensor cuda FAILED", repr(e)[:200], flush=True)
    try:
        t = torch.tensor([rank + 1.0], device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print(rank, "all_reduce cuda ok", t.item(), flush=True)
        dist.barrier(); print(rank, "barrier ok", flush=True)
        lst = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(lst, torch.tensor([rank + 0.5], device=dev, dtype=torch.float64)); print(rank, "all_gather list cuda ok", [x.item() for x in lst], flush=True)
    except Exception as e:
        print(rank, "other FAILED", repr(e)[:200], flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2, 29533), nprocs=2, join=True)
