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
import ctypes as C

import torch

from . import _lib
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
            if self.device.type == "cuda":                              # cumulative: t[r] on indices (ontovae_compose_order)
                rd = r.to(self.device)
                new = torch.empty_like(self.order)
                with torch.cuda.device(self.device):
                    _lib.check(_lib.lib().ontovae_compose_order(C.c_void_p(self.order.data_ptr()), C.c_void_p(rd.data_ptr()),
                                                                C.c_void_p(new.data_ptr()), self.dataset_len, self._stream()),
                               "compose_order")
                self.order = new
            else:
                self.order = self.order[r].contiguous()
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
        return tuple(self._gather(t, idx) for t in self.tensors)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _gather(self, t, idx):
        """out[b, :] = t[idx[b], :]: ontovae_gather_rows for float32 matrices in HBM, torch indexing for anything else."""
        if self.device.type != "cuda" or t.dtype != torch.float32 or t.dim() != 2 or not t.is_contiguous() or t.shape[1] < 1:
            return t.index_select(0, idx.long())
        out = torch.empty(idx.shape[0], t.shape[1], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ontovae_gather_rows(C.c_void_p(t.data_ptr()), t.stride(0), C.c_void_p(idx.data_ptr()), idx.shape[0],
                                                      t.shape[1], C.c_void_p(out.data_ptr()), self._stream()), "gather_rows")
        return out

    def __len__(self):
        return self.n_batches
