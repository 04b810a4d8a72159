"""Latency of FlatGradSync's host-side presence exchange (one uint8 per parameter, MAX all-reduce over gloo) at world sizes 2 and 8, on
this machine's CPUs over loopback: python tools/r5/gloo_latency.py"""
import os, socket, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(158, dtype=torch.uint8)                   # one byte per trainable parameter tensor of DGCNN
    for _ in range(50):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ts = []
    for _ in range(300):
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    if rank == 0:
        q.put((world, 1e6 * ts[len(ts) // 2], 1e6 * ts[int(0.95 * len(ts))]))
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    for world in (2, 8):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in ps]
        w, med, p95 = q.get(timeout=300)
        [p.join() for p in ps]
        print("gloo MAX all-reduce of 158 bytes, world %d (%d CPUs here): median %.0f us, p95 %.0f us" % (w, os.cpu_count(), med, p95))
