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
import os

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


class _DpState:
    """Per (engine, global batch, world, rank) buffers and index maps of the data-parallel step, built once."""

    def __init__(self, e, B, world, rank):
        L, dev = e.plan.L, e.device
        self.start, self.size = shard_bounds(B, world, rank)
        if self.size < 1:
            raise ValueError(f"global batch {B} leaves rank {rank} of {world} without rows")
        sizes = [shard_bounds(B, world, r)[1] for r in range(world)]
        starts = [shard_bounds(B, world, r)[0] for r in range(world)]
        self.stats_send = torch.zeros(2 * L + 1, device=dev)          # [mean | M2 | rows of this rank]
        self.stats_send[2 * L] = float(self.size)
        self.stats = self.stats_send[:2 * L].view(2, L)
        self.stats_all = torch.zeros(world * (2 * L + 1), device=dev)      # flat: gloo's all_gather_into_tensor wants 1-D
        self.counts = torch.tensor([float(n) for n in sizes], device=dev)
        self.gstats = torch.zeros(2, L, device=dev)
        self.bstats = torch.zeros(2, L, device=dev)
        self.lloc = torch.zeros(3, device=dev)
        # all-gathered dh: per rank [L][blm] + [L] (local b0 gradient), stride padded to 4 floats
        self.blm = (max(sizes) + 3) // 4 * 4
        self.rstride = (L * self.blm + L + 3) // 4 * 4
        self.send = torch.zeros(self.rstride, device=dev)
        self.gathered = torch.zeros(world * self.rstride, device=dev)
        # position k = r * blm + j of the gathered columns -> position in the global batch (padding: the rank's first row)
        m = []
        for r in range(world):
            m += [starts[r] + j if j < sizes[r] else starts[r] for j in range(self.blm)]
        self.colmap = torch.tensor(m, dtype=torch.int64, device=dev)
        self.colmap32 = self.colmap.to(torch.int32)


def _dp_state(e, B, world, rank):
    cache = getattr(e, "_dp_cache", None)
    if cache is None:
        cache = e._dp_cache = {}
    key = (int(B), int(world), int(rank))
    if key not in cache:
        cache[key] = _DpState(e, B, world, rank)
    return cache[key]


def dp_train_step(model, x, xidx, eps, kl_coeff, opt, loss_out, group=None, reduce_loss=True):
    """One synchronous data-parallel step.  `x`/`xidx` address the GLOBAL batch (identical on all ranks); `eps`, if
    given, is the global [B, L] draw.  Exchanges: SyncBN forward (all-gather of 2L+1 floats), SyncBN backward
    (all-reduce of 2L floats), gradients of the heads and the decoder (all-reduce, 3 MB at GO-BP scale, overlapped
    with the BatchNorm backward and dW0), and dh [L, B_local] (all-gather): dW0 = dh x of the GLOBAL batch is then
    formed, and applied, by every rank itself (ontovae_dw0_global), so the 28.8 MB gradient of the first layer never
    crosses the links.  ONTOVAE_DP_ALLREDUCE=1 selects the plain all-reduce of the whole gradient buffer instead.
    reduce_loss=False leaves this rank's partial sums in loss_out (the caller all-reduces a whole epoch at once)."""
    if os.environ.get("ONTOVAE_DP_ALLREDUCE", "0") == "1":
        return dp_train_step_allreduce(model, x, xidx, eps, kl_coeff, opt, loss_out, group, reduce_loss)
    e = model._engine
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = x.shape[0] if xidx is None else xidx.shape[0]
    st = _dp_state(e, B, world, rank)
    start, size, L = st.start, st.size, e.plan.L
    if xidx is None:
        xl, il = x[start:start + size], None
    else:
        if xidx.dtype != torch.int32 or xidx.device != e.device:
            xidx = xidx.to(device=e.device, dtype=torch.int32)
        xl, il = x, xidx[start:start + size]
    el = None if eps is None else torch.as_tensor(eps)[start:start + size]
    off = e.next_offset() * world + rank          # distinct Philox offsets per rank
    lay = e.plan.layout

    lbuf = loss_out if not reduce_loss else st.lloc        # phase 1 writes {rec, kl, rec + kl_coeff kl} of this rank's rows
    e.train_step_dp(0, xl, il, B, el, kl_coeff, st.stats, st.bstats, lbuf, off)
    dist.all_gather_into_tensor(st.stats_all, st.stats_send, group=group)

    e.train_step_dp(16 + world, xl, il, B, el, kl_coeff, st.stats_all, st.bstats, lbuf, off)
    dist.all_reduce(st.bstats, group=group)

    e.dp_bn_bwd(size, B, st.bstats, off, st.send, st.blm)
    dist.all_gather_into_tensor(st.gathered, st.send, group=group)
    # issued after the all-gather (collectives of one communicator run in order, and dW0 waits for the gather only);
    # it then runs under the dW0 kernel
    h_early = dist.all_reduce(e.Gr[int(lay.Wh):int(lay.total)], group=group, async_op=True)
    hyper = opt.next_hyper()
    if xidx is None:
        e.dw0_global(e._x(x, None)[0], None, st.colmap32, world, st.blm, st.gathered, st.rstride, hyper)
    else:
        e.dw0_global(x, xidx, st.colmap32, world, st.blm, st.gathered, st.rstride, hyper)
    h_early.wait()
    e.adamw_apply_range(hyper, int(lay.b0), int(lay.total))
    if reduce_loss:
        dist.all_reduce(st.lloc, group=group)
        loss_out.copy_(st.lloc)


def dp_train_step_allreduce(model, x, xidx, eps, kl_coeff, opt, loss_out, group=None, reduce_loss=True):
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
    if reduce_loss:
        dist.all_reduce(lloc, group=group)
    h_early.wait()
    h_late.wait()
    e.adamw_apply(opt.next_hyper())
    loss_out[0] = lloc[0]
    loss_out[1] = lloc[1]
    loss_out[2] = lloc[0] + float(kl_coeff) * lloc[1]
