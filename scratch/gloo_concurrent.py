"""CPU-only: the collective pattern of the all-reduce form (3.2 MB + 4 x 7.7 MB + 4 KB async SUM all-reduces outstanding
together, then a small synchronous all-gather) on plain CPU tensors over gloo, two processes.  Does gloo itself stall?"""
import os, sys, time, faulthandler
import torch, torch.distributed as dist
rank, world, port, iters = int(sys.argv[1]), 2, sys.argv[2], int(sys.argv[3])
faulthandler.dump_traceback_later(60, exit=True)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=world)
bufs = [torch.ones(809600)] + [torch.ones(1920000) for _ in range(3)] + [torch.ones(1440000), torch.ones(1080)]
small, out = torch.ones(721), torch.empty(2 * 721)
t0 = time.time()
for it in range(iters):
    dist.all_gather_into_tensor(out, small)
    dist.all_reduce(torch.ones(720))
    hs = [dist.all_reduce(b, async_op=True) for b in bufs]
    for h in hs:
        h.wait()
    for b in bufs:
        b.fill_(1.0)
print(rank, "ok", iters, "iterations", round(time.time() - t0, 2), "s", flush=True)
dist.destroy_process_group()
