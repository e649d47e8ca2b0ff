"""Data-parallel glue on CPU with gloo, world_size 2: row sharding, the SyncBN statistics
combine and the bucketed SUM all-reduce reproduce the single-process global-batch step.  The
compute legs are the CPU oracle's own formulas (the HIP legs are covered on the GPU by
test_dp_phases_equal_fused_step); what is under test here is onto_vae_amd/parallel.py."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from onto_vae_amd.parallel import shard_bounds, combine_bn_stats, grad_buckets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(0)
    B, L, P = 37, 6, 1000
    h = torch.tensor(rng.randn(B, L) * 3 + 1.5, dtype=torch.float32)      # global pre-BN activations
    gfull = torch.tensor(rng.randn(B, P), dtype=torch.float32)           # per-sample gradient contributions
    start, size = shard_bounds(B, world, rank)
    hl = h[start:start + size]
    # SyncBN forward: all-gather of (mean, M2, n), Chan combine
    mean_l = hl.mean(0)
    m2_l = ((hl - mean_l) ** 2).sum(0)
    packed = torch.cat([mean_l, m2_l, torch.tensor([float(size)])])
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed)
    g = torch.stack(gathered)
    stats = combine_bn_stats(g[:, :2 * L].view(world, 2, L), g[:, 2 * L])
    # gradients: SUM all-reduce in the two buckets of the flat buffer
    class Lay:
        Wh, total = 400, P
    grads = gfull[start:start + size].sum(0)
    late, early = grad_buckets(Lay)
    assert late == (0, 400) and early == (400, 1000)
    h1 = dist.all_reduce(grads[early[0]:early[1]], async_op=True)
    h2 = dist.all_reduce(grads[late[0]:late[1]], async_op=True)
    h1.wait(); h2.wait()
    if rank == 0:
        out["stats"] = stats.numpy()
        out["grads"] = grads.numpy()
        out["want_mean"] = h.mean(0).numpy()
        out["want_var"] = h.var(0, unbiased=False).numpy()
        out["want_grads"] = gfull.sum(0).numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_dp_collectives_world2():
    from onto_vae_amd.parallel import shard_bounds
    # shards tile the batch, sizes differ by at most one, no rank is empty when B >= world
    for B in (2, 3, 37, 128, 8192):
        for world in (1, 2, 4, 8):
            if B < world:
                continue
            parts = [shard_bounds(B, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(s for _, s in parts) == B
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(s for _, s in parts) - min(s for _, s in parts) <= 1
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    np.testing.assert_allclose(out["stats"][0], out["want_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["stats"][1], out["want_var"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["grads"], out["want_grads"], rtol=1e-5, atol=1e-5)
