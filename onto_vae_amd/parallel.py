"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI) for the three exchanges a step needs.  The reference is single-device (SURVEY.md section 2
row 12); the contract is that an R-rank step equals the single-device step on the global batch:

  * every rank sees the same global batch order and takes rows shard_bounds(B, R, r) of it;
  * SyncBN forward: all-gather of per-rank (mean, M2, n), combined with Chan's formula;
  * SyncBN backward: all-reduce SUM of (sum dy, sum dy*xhat);
  * gradients: all-reduce SUM (the loss is a SUM over the batch, reference vae_model.py:124-129),
    in two buckets of the flat gradient buffer: [heads | decoder] is complete after phase 1 and
    its all-reduce overlaps the dW0 GEMM of phase 2; [W0 | b0 | gamma | beta] follows.

No collective is on the inference path: samples are independent, ranks take disjoint slices.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world, rank):
    """Rows [start, start+size) of a global batch of B rows owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(B), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def combine_bn_stats(stats, counts):
    """stats [R, 2, L] = per-rank (mean, M2 = sum (h - mean_r)^2), counts [R] -> [2, L] =
    (global mean, global biased variance)."""
    counts = counts.to(stats.dtype).view(-1, 1)
    n = counts.sum()
    mean = (stats[:, 0] * counts).sum(0) / n
    m2 = (stats[:, 1] + counts * (stats[:, 0] - mean) ** 2).sum(0)
    return torch.stack([mean, m2 / n])


def grad_buckets(layout):
    """(late, early) float ranges of the flat gradient buffer: `early` (heads + decoder) is ready
    after phase 1, `late` (first encoder layer) after phase 2."""
    return (0, int(layout.Wh)), (int(layout.Wh), int(layout.total))


def dp_train_step(model, x, xidx, eps, kl_coeff, opt, loss_out, group=None):
    """One synchronous data-parallel step.  `x`/`xidx` address the GLOBAL batch (identical on all
    ranks); `eps`, if given, is the global [B, L] draw."""
    e = model._engine
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = x.shape[0] if xidx is None else xidx.shape[0]
    start, size = shard_bounds(B, world, rank)
    if size < 1:
        raise ValueError(f"global batch {B} leaves rank {rank} of {world} without rows")
    if xidx is None:
        xl, il = x[start:start + size], None
    else:
        xl, il = x, xidx[start:start + size]
    el = None if eps is None else torch.as_tensor(eps)[start:start + size]
    L = e.plan.L
    dev = e.device
    stats = torch.zeros(2, L, device=dev)
    bstats = torch.zeros(2, L, device=dev)
    lloc = torch.zeros(3, device=dev)
    off = e.next_offset() * world + rank          # distinct Philox offsets per rank

    e.train_step_dp(0, xl, il, B, el, kl_coeff, stats, bstats, lloc, off)
    gathered = [torch.empty(2 * L + 1, device=dev) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([stats.flatten(), torch.tensor([float(size)], device=dev)]), group=group)
    g = torch.stack(gathered)
    stats.copy_(combine_bn_stats(g[:, :2 * L].view(world, 2, L), g[:, 2 * L]))

    e.train_step_dp(1, xl, il, B, el, kl_coeff, stats, bstats, lloc, off)
    dist.all_reduce(bstats, group=group)
    late, early = grad_buckets(e.plan.layout)
    h_early = dist.all_reduce(e.Gr[early[0]:early[1]], group=group, async_op=True)

    e.train_step_dp(2, xl, il, B, el, kl_coeff, stats, bstats, lloc, off)
    h_late = dist.all_reduce(e.Gr[late[0]:late[1]], group=group, async_op=True)
    dist.all_reduce(lloc, group=group)
    h_early.wait()
    h_late.wait()
    e.adamw_apply(opt.next_hyper())
    loss_out[0] = lloc[0]
    loss_out[1] = lloc[1]
    loss_out[2] = lloc[0] + float(kl_coeff) * lloc[1]
