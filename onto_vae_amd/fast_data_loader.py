"""HBM-resident batcher with the interface of reference onto_vae/fast_data_loader.py.

Same constructor, same iteration protocol (tuples of slices, last partial batch kept, cumulative
shuffle: every `iter()` applies a fresh `torch.randperm(N)` on top of the current order, drawn
from torch's global CPU generator exactly like the reference's line 33, so a given
`torch.manual_seed` yields the reference's batch order).

What changes: the tensors live in HBM once (moved to `device` at construction) and are never
permuted.  The shuffle only composes a device-side int32 `order` array
(order <- order[r], reference line 34 applied to indices), and the training loop reads a batch
as (dataset, order[i:i+B]): the kernels gather the rows themselves, so there is no per-epoch
N x G copy and no per-step host-to-device copy (reference lines 34 and 41,
onto_vae/vae_model.py:156).
"""
import torch

from .engine import norm_device


class FastTensorDataLoader:
    def __init__(self, *tensors, batch_size=32, shuffle=False, device=None):
        assert all(t.shape[0] == tensors[0].shape[0] for t in tensors)
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = norm_device(device)
        self.tensors = tuple(t.to(self.device) for t in tensors)
        self.dataset_len = self.tensors[0].shape[0]
        self.batch_size = batch_size
        self.shuffle = shuffle
        n_batches, remainder = divmod(self.dataset_len, self.batch_size)
        if remainder > 0:
            n_batches += 1
        self.n_batches = n_batches
        self.order = torch.arange(self.dataset_len, dtype=torch.int32, device=self.device)
        self.i = 0

    def __iter__(self):
        if self.shuffle:
            r = torch.randperm(self.dataset_len)                       # CPU generator, as the reference
            self.order = self.order[r.to(self.device)].contiguous()     # cumulative: t[r] on indices
        self.i = 0
        return self

    def next_indices(self):
        """Row indices (int32, device) of the next batch, or None at the end of the epoch."""
        if self.i >= self.dataset_len:
            return None
        idx = self.order[self.i:self.i + self.batch_size]
        self.i += self.batch_size
        return idx

    def __next__(self):
        idx = self.next_indices()
        if idx is None:
            raise StopIteration
        li = idx.long()
        return tuple(t.index_select(0, li) for t in self.tensors)

    def __len__(self):
        return self.n_batches
