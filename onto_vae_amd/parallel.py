"""Data-parallel training: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI) for the three exchanges a step needs.  The reference is single-device (SURVEY.md section 2
row 12); the contract is that an R-rank step equals the single-device step on the global batch:

  * every rank sees the same global batch order and takes rows shard_bounds(B, R, r) of it;
  * SyncBN forward: all-gather of per-rank (mean, M2, n), combined with Chan's formula;
  * SyncBN backward + first layer: ONE all-gather of [dy | xhat | (sum dy, sum dy*xhat)] per rank; every rank adds
    the sums in rank order, forms dh of all columns and dW0 = dh x of the global batch itself (small global batches),
    or: all-reduce SUM of the sums, then all-reduce of the first layer's gradient (large global batches);
  * the other gradients: all-reduce SUM (the loss is a SUM over the batch, reference vae_model.py:124-129) of the
    [heads | decoder] part of the flat gradient buffer, complete after phase 1, under the dW0 kernel.

No collective is on the inference path: samples are independent, ranks take disjoint slices.
"""
import os
import queue
import threading

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        pass


def _stream_ordered(group):
    """nccl (RCCL) collectives are ordered on streams and may be left outstanding.  Any other backend (gloo: the CPU and
    2-process tests) reaches device buffers through _HostStaged below."""
    return dist.get_backend(group) == "nccl"


class _Staged:
    """Handle of one collective of _HostStaged.  wait(): the caller's stream is ordered after the copy back to the device
    (as RCCL's Work.wait() orders it after the collective); the host blocks only until the worker has enqueued that copy."""

    def __init__(self):
        self.done = threading.Event()
        self.back = None
        self.error = None

    def wait(self):
        self.done.wait()
        if self.error is not None:
            raise self.error
        if self.back is not None:
            torch.cuda.current_stream(self.back_device).wait_event(self.back)


class _HostStaged:
    """RCCL's calling convention on top of a host backend (gloo), for device tensors: a collective is issued against what is
    on the compute stream at the moment of the call, runs in the background while the caller keeps launching kernels, and is
    complete for the caller's stream after wait().  One worker thread per process takes the calls in issue order (the order is
    the same on every rank, as a communicator's order is): it waits for the issue-time event on a copy stream, copies the
    buffer into a pinned mirror, runs the collective on the mirror, copies the result back in place and records an event.

    Why not hand gloo the device tensors: its own staging of them stalled with two processes on one GPU (DESIGN.md section 5
    has what is known); and staging every collective synchronously, as round 2 did, never left a handle outstanding under
    later launches, so ordering and buffer-reuse errors of the panel loop could not show before an 8-GPU run."""

    _inst = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        if key not in cls._inst:
            cls._inst[key] = cls(device)
        return cls._inst[key]

    def __init__(self, device):
        self.device = device
        self.q = queue.Queue()
        self.copy_stream = torch.cuda.Stream(device)
        self.mirrors = {}                       # numel -> free pinned float buffers (allocated on the caller's thread)
        self.thread = threading.Thread(target=self._run, name="ontovae-host-staged", daemon=True)
        self.thread.start()

    def _mirror(self, t):
        free = self.mirrors.setdefault((t.numel(), t.dtype), [])
        return free.pop() if free else torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)

    def submit(self, kind, t, out, group):
        h = _Staged()
        h.back_device = self.device
        issued = torch.cuda.Event()
        issued.record(torch.cuda.current_stream(self.device))
        h.back = torch.cuda.Event()
        self.q.put((kind, t, out, group, issued, self._mirror(t), self._mirror(out) if out is not None else None, h))
        return h

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            kind, t, out, group, issued, ht, ho, h = self.q.get()
            try:
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(issued)
                    ht.copy_(t.detach().reshape(-1), non_blocking=True)
                    self.copy_stream.synchronize()
                    if kind == "all_reduce":
                        dist.all_reduce(ht, group=group)
                        t.detach().reshape(-1).copy_(ht, non_blocking=True)
                    else:
                        dist.all_gather_into_tensor(ho, ht, group=group)
                        out.detach().reshape(-1).copy_(ho, non_blocking=True)
                    h.back.record(self.copy_stream)
                    # the mirrors may be reused once the copy back has left them
                    self.copy_stream.synchronize()
                self.mirrors[(ht.numel(), ht.dtype)].append(ht)
                if ho is not None:
                    self.mirrors[(ho.numel(), ho.dtype)].append(ho)
            except BaseException as e:          # surfaced by wait()
                h.error = e
            h.done.set()


def _direct(group):
    """ONTOVAE_DP_GLOO_DEVICE=1 (diagnosis only, profiling/dp_hang_probe.sh): hand the host backend the device tensors."""
    return _stream_ordered(group) or os.environ.get("ONTOVAE_DP_GLOO_DEVICE") == "1"


def all_reduce(t, group=None, async_op=False):
    """SUM all-reduce of `t` in place.  Device tensors must be contiguous.  Returns a handle with wait()."""
    if _direct(group):
        return dist.all_reduce(t, group=group, async_op=async_op) or _Done()
    if not t.is_cuda:
        return dist.all_reduce(t, group=group, async_op=async_op) or _Done()
    if not t.is_contiguous():
        raise ValueError("all_reduce of a device tensor needs a contiguous tensor")
    h = _HostStaged.get(t.device).submit("all_reduce", t, None, group)
    if not async_op:
        h.wait()
        return _Done()
    return h


def all_gather_into_tensor(out, t, group=None):
    if _direct(group) or not t.is_cuda:
        dist.all_gather_into_tensor(out, t, group=group)
        return
    _HostStaged.get(t.device).submit("all_gather", t, out, group).wait()


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
        # one exchange instead of two (ontovae_dp_pack_dy / ontovae_dp_dh_global): per rank [L][blm] dy | [L] | [L][blm] xhat | [2 L]
        self.sizes = sizes
        self.rstride2 = (2 * L * self.blm + 3 * L + 3) // 4 * 4
        self.send2 = torch.zeros(self.rstride2, device=dev)
        self.gathered2 = torch.zeros(world * self.rstride2, device=dev)


def w0_row_panels(L, world, n_panels=None):
    """Row ranges of W0 for the panel-wise all-reduce: n panels of (almost) equal size, multiples of 32 rows (the tile height
    of the dW0 kernel) except the last.  ONTOVAE_DP_W0_PANELS overrides the count (1 = one all-reduce after the product)."""
    if n_panels is None:
        n_panels = int(os.environ.get("ONTOVAE_DP_W0_PANELS", "4"))
    n_panels = max(1, min(n_panels, (L + 31) // 32))
    per = ((L + n_panels - 1) // n_panels + 31) // 32 * 32
    return [(l0, min(l0 + per, L)) for l0 in range(0, L, per)]


def _dp_state(e, B, world, rank):
    cache = getattr(e, "_dp_cache", None)
    if cache is None:
        cache = e._dp_cache = {}
    key = (int(B), int(world), int(rank))
    if key not in cache:
        cache[key] = _DpState(e, B, world, rank)
    return cache[key]


# all-reduce time of the first layer's gradient (28.8 MB at GO-BP scale) on point-to-point xGMI, microseconds per MB:
# 2 ranks share one link, 4 ranks three, 8 ranks seven (~64 GB/s per direction each); ring / direct all-reduce moves
# 2 (R-1)/R of the buffer per rank
def _allreduce_us(nbytes, world):
    links = max(world - 1, 1)
    return 20.0 + 2.0 * (world - 1) / world * nbytes / (links * 64e9) * 1e6


def use_gathered_dh(B_global, world, L, G):
    """Which exchange for the first layer: every rank forming dW0 of the global batch from all-gathered dh costs MFMA
    time proportional to the global batch (measured: 0.155 us per row at L = 360, G = 20000, i.e. 2.15e-8 us per row
    and weight); all-reducing the gradient costs the link time above plus the unfused AdamW pass (~35 us per 7.2 M
    weights).  ONTOVAE_DP_ALLREDUCE=0/1 overrides."""
    env = os.environ.get("ONTOVAE_DP_ALLREDUCE")
    if env in ("0", "1"):
        return env == "0"
    w = float(L) * float(G)
    extra_mfma = 2.15e-8 * w * B_global * (1.0 - 1.0 / world)
    return extra_mfma < _allreduce_us(4.0 * w, world) + 35.0 * w / 7.2e6


def dp_train_step(model, x, xidx, eps, kl_coeff, opt, loss_out, group=None, reduce_loss=True):
    """One synchronous data-parallel step.  `x`/`xidx` address the GLOBAL batch (identical on all ranks); `eps`, if
    given, is the global [B, L] draw.  Exchanges: SyncBN forward (all-gather of 2L+1 floats per rank, combined inside
    the BatchNorm kernel), SyncBN backward (all-reduce of 2L floats), gradients of the heads and the decoder
    (all-reduce, 3 MB at GO-BP scale, under the dW0 kernel), and for the first layer either
      * an all-gather of [dy | xhat | local sums] (which also replaces the SyncBN-backward all-reduce): every rank
        then forms dh of all columns and dW0 = dh x of the GLOBAL batch itself, and applies it (ontovae_dp_dh_global,
        ontovae_dw0_global), so the 28.8 MB gradient never crosses the links; or
      * the all-reduce of that gradient followed by the flat AdamW pass,
    whichever use_gathered_dh estimates to be shorter (small global batches: the first; large ones: the second).
    reduce_loss=False leaves this rank's partial sums in loss_out (the caller all-reduces a whole epoch at once)."""
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
    all_gather_into_tensor(st.stats_all, st.stats_send, group=group)

    e.train_step_dp(16 + world, xl, il, B, el, kl_coeff, st.stats_all, st.bstats, lbuf, off)

    hyper = opt.next_hyper()
    if use_gathered_dh(B, world, L, e.plan.G):
        if os.environ.get("ONTOVAE_DP_TWO_EXCHANGES") == "1":
            # all-reduce of the BatchNorm-backward sums, then all-gather of dh
            all_reduce(st.bstats, group=group)
            e.dp_bn_bwd(size, B, st.bstats, off, st.send, st.blm)
            all_gather_into_tensor(st.gathered, st.send, group=group)
            gathered, rstride = st.gathered, st.rstride
        else:
            # one exchange: all-gather of [dy | xhat | local sums]; every rank adds the sums in rank order and forms dh of
            # all columns itself (one small collective less on the critical path)
            e.dp_pack_dy(size, st.send2, st.blm)
            all_gather_into_tensor(st.gathered2, st.send2, group=group)
            e.dp_dh_global(B, st.sizes, st.blm, st.gathered2, st.rstride2)
            gathered, rstride = st.gathered2, st.rstride2
        # issued after the all-gather (collectives of one communicator run in order, and dW0 waits for the gather
        # only); it then runs under the dW0 kernel
        h_early = all_reduce(e.Gr[int(lay.Wh):int(lay.total)], group=group, async_op=True)
        if xidx is None:
            e.dw0_global(e._x(x, None)[0], None, st.colmap32, world, st.blm, gathered, rstride, hyper)
        else:
            e.dw0_global(x, xidx, st.colmap32, world, st.blm, gathered, rstride, hyper)
        h_early.wait()
        e.adamw_apply_range(hyper, int(lay.b0), int(lay.total))
    else:
        all_reduce(st.bstats, group=group)
        late, early = grad_buckets(lay)
        h_early = all_reduce(e.Gr[early[0]:early[1]], group=group, async_op=True)     # under the dW0 products
        # W0's gradient (28.8 MB at GO-BP scale, the bulk) in contiguous ROW panels: panel p's all-reduce is issued right
        # after its product and runs under the products of the panels after it (the collective's stream waits only for what
        # is on the compute stream when it is issued); what stays exposed is the last panel's exchange
        e.train_step_dp(3, xl, il, B, el, kl_coeff, st.stats, st.bstats, lbuf, off)        # BatchNorm backward -> dh, small gradients
        G = e.plan.G
        handles = []
        for l0, l1 in w0_row_panels(L, world):
            e.dw0_rows(xl, il, l0, l1)
            handles.append((l0, l1, all_reduce(e.Gr[int(lay.W0) + l0 * G:int(lay.W0) + l1 * G], group=group, async_op=True)))
        h_small = all_reduce(e.Gr[int(lay.b0):late[1]], group=group, async_op=True)   # b0, gamma, beta
        for l0, l1, h in handles:
            h.wait()
            e.adamw_apply_range(hyper, int(lay.W0) + l0 * G, int(lay.W0) + l1 * G)
        h_small.wait()
        h_early.wait()
        e.adamw_apply_range(hyper, int(lay.b0), int(lay.total))
    if reduce_loss:
        all_reduce(st.lloc, group=group)
        loss_out.copy_(st.lloc)
