"""Device state + launch layer between the Python OntoVAE surface and the C ABI.

Owns the flat parameter buffer, the packed decoder structure on the device, the workspace and
the ontovae_ctx_t that every C entry point takes.  torch is plumbing here: it allocates HBM,
provides the stream and (for data parallel) the collectives.
"""
import ctypes as C
import functools
import math

import numpy as np
import torch

from . import _lib
from .packing import MaskPlan

ADAM_DEFAULTS = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)   # torch.optim.AdamW defaults


def norm_device(device):
    """torch.device with an explicit index for GPUs, so that device comparisons are meaningful."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _on_device(fn):
    """Run an entry point with the engine's GPU as the current HIP device: the library keys its per-device state (LDS
    attributes, side stream) on hipGetDevice(), and a launch into another device's stream fails."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if self.device.type != "cuda" or torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **k)
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, plan: MaskPlan, device, drop=0.0, z_drop=0.0):
        self.plan = plan
        self.device = norm_device(device)
        self.drop = float(drop)
        self.z_drop = float(z_drop)
        lay = plan.layout
        self.P = torch.zeros(lay.total, dtype=torch.float32, device=self.device)
        self.M = None
        self.V = None
        self.Gr = None
        self.rmean = torch.zeros(plan.L, dtype=torch.float32, device=self.device)
        self.rvar = torch.ones(plan.L, dtype=torch.float32, device=self.device)
        self.nbt = torch.zeros((), dtype=torch.int64, device=self.device)
        self._ws = None
        self.max_batch = 0
        self.max_train_batch = 0       # rows a training pass may take: the training-only buffers of the workspace are sized by it
        self._ctx = None
        self._idx = None
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.rng_offset = 0
        self.flags = 0          # _lib FLAG_* bits: force the general-purpose kernel variants (tests), bf16 variant
        self._bf16_ws = None
        self._upload_plan()

    # ------------------------------------------------------------------ views of the flat buffer
    def view(self, name, buf=None):
        p, lay = self.plan, self.plan.layout
        buf = self.P if buf is None else buf
        L, G = p.L, p.G
        if name == "W0":
            return buf[lay.W0:lay.W0 + L * G].view(L, G)
        if name in ("b0", "gamma", "beta"):
            o = getattr(lay, name)
            return buf[o:o + L]
        if name == "Wmu":
            return buf[lay.Wh:lay.Wh + L * L].view(L, L)
        if name == "Wlv":
            return buf[lay.Wh + L * L:lay.Wh + 2 * L * L].view(L, L)
        if name == "bmu":
            return buf[lay.bh:lay.bh + L]
        if name == "blv":
            return buf[lay.bh + L:lay.bh + 2 * L]
        if name == "dec_vals":
            return buf[lay.dec_vals:lay.dec_vals + p.nnz]
        if name == "dec_bias":
            return buf[lay.dec_bias:lay.dec_bias + p.C + p.G]
        raise KeyError(name)

    def layer_vals(self, i, buf=None):
        k0, k1 = self.plan.layer_entries[i]
        return self.view("dec_vals", buf)[k0:k1]

    def layer_bias(self, i, buf=None):
        n0, no = self.plan.layer_nodes[i]
        return self.view("dec_bias", buf)[n0:n0 + no]

    def layer_dense(self, i, buf=None):
        """Dense [out_i, in_i] weight of decoder layer i (the reference's state_dict tensor)."""
        cout, cin = self.plan.layer_shapes[i]
        k0, k1 = self.plan.layer_entries[i]
        dense = torch.zeros(cout, cin, dtype=torch.float32, device=self.device)
        if k1 > k0 and self.device.type != "cuda":      # state_dict export without the GPU (host plumbing only)
            er, ec = self._entry_rc()
            dense[er[k0:k1].long(), ec[k0:k1].long()] = self.layer_vals(i, buf)
        elif k1 > k0:
            self._scatter_vals(self.view("dec_vals", buf), k0, k1, dense)
        return dense

    def _scatter_vals(self, vals, k0, k1, dense):
        """dense[row, col of entry k] = vals[k], k in [k0, k1) (ontovae_vals_to_dense)."""
        self._require_gpu()
        er, ec = self._entry_rc()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ontovae_vals_to_dense(_ptr(vals), _ptr(er), _ptr(ec), int(k0), int(k1), _ptr(dense), dense.stride(0),
                                                        self.stream()), "vals_to_dense")

    def set_layer_dense(self, i, dense, buf=None):
        """Import a dense weight; returns the largest |value| found off the mask (dropped)."""
        cout, cin = self.plan.layer_shapes[i]
        dense = torch.as_tensor(dense, dtype=torch.float32, device=self.device)
        if tuple(dense.shape) != (cout, cin):
            raise ValueError(f"decoder layer {i}: expected shape {(cout, cin)}, got {tuple(dense.shape)}")
        dense = dense.contiguous()
        k0, k1 = self.plan.layer_entries[i]
        if self.device.type != "cuda":      # a model can be constructed, loaded and saved without the GPU
            er, ec = self._entry_rc()
            r, c = er[k0:k1].long(), ec[k0:k1].long()
            self.layer_vals(i, buf).copy_(dense[r, c])
            rest = dense.clone()
            rest[r, c] = 0
            return float(rest.abs().max()) if rest.numel() else 0.0
        er, ec = self._entry_rc()
        vals = self.view("dec_vals", buf)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ontovae_dense_to_vals(_ptr(dense), dense.stride(0), _ptr(er), _ptr(ec), int(k0), int(k1), _ptr(vals),
                                                        self.stream()), "dense_to_vals")
        on_mask = torch.zeros_like(dense)
        if k1 > k0:
            self._scatter_vals(vals, k0, k1, on_mask)
        return float((dense - on_mask).abs().max()) if dense.numel() else 0.0

    def _entry_rc(self):
        if self._idx is None or "er" not in self._idx:
            self._upload_plan()
        return self._idx["er"], self._idx["ec"]

    # ------------------------------------------------------------------ device plumbing
    def _upload_plan(self):
        p = self.plan
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self._sell = []
        for sd in p.sell:
            dv = {f: t(sd[f]) for f in ("node", "info", "chunk0", "idx", "k")}
            dv["w"] = torch.zeros(max(sd["n_chunks"], 1) * 512, dtype=torch.float32, device=dev)
            dv["bias"] = torch.zeros(max(sd["n_slices"], 1) * 64, dtype=torch.float32, device=dev)
            self._sell.append(dv)
        gs = p.gsell
        self._gsell = {f: t(gs[f].view(np.int16) if f == "col" else gs[f]) for f in ("gene", "quad0", "col", "k")}
        self._gsell["w"] = torch.zeros(max(gs["n_quads"], 1) * 256, dtype=torch.float32, device=dev)
        self._idx = dict(rowptr=t(p.rowptr), colidx=t(p.colidx), cscptr=t(p.cscptr), csc_row=t(p.csc_row),
                         csc_k=t(p.csc_k), csc_split=t(p.csc_split), row_k=t(p.row_k), er=t(p.entry_row), ec=t(p.entry_col),
                         gene_tiles=t(p.gene_tiles), panel_tiles=t(p.panel_tiles), term_tiles=t(p.term_tiles),
                         tpanel_tiles=t(p.tpanel_tiles) if p.tpanel_tiles is not None else None)
        self._ctx = None

    def to(self, device):
        device = norm_device(device)
        if device == self.device:
            return self
        self.device = device
        for n in ("P", "M", "V", "Gr", "rmean", "rvar", "nbt"):
            v = getattr(self, n)
            if v is not None:
                setattr(self, n, v.to(device))
        self._ws = None
        self.max_batch = 0
        self.max_train_batch = 0
        self._upload_plan()
        return self

    def ensure_opt_state(self):
        if self.M is None:
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)
            self._ctx = None

    def ensure_grad_buffer(self):
        if self.Gr is None:
            self.Gr = torch.zeros_like(self.P)
            self._ctx = None

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise _lib.OntoVAEHipError(
                "the OntoVAE hot path runs on an MI355X through libontovae_hip.so; there is no CPU path "
                f"(model is on {self.device})")

    def ctx(self, batch, train=False):
        """The ontovae_ctx_t for a launch with `batch` rows (grows the workspace when needed).  train: a training pass, which
        also uses the training-only buffers (weight-gradient partials: batch / 32 copies of the packed decoder); those are
        sized by the largest TRAINING batch seen, not by the inference batch."""
        self._require_gpu()
        L = _lib.lib()
        if batch > self.max_batch or (train and batch > self.max_train_batch):
            mb = max(int(batch), self.max_batch, 128)
            mtb = max(int(batch) if train else 0, self.max_train_batch, 128)
            nbytes = L.ontovae_workspace_bytes_split(C.byref(self.plan.dims), mb, mtb)
            if nbytes <= 0:
                raise _lib.OntoVAEHipError("workspace sizing failed")
            self._ws = None
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            if mb > self.max_batch:
                self._bf16_ws = None
            self.max_batch, self.max_train_batch = mb, mtb
            self._ctx = None
        if (self.flags & _lib.FLAG_BF16_GEMM) and getattr(self, "_bf16_ws", None) is None:
            nb16 = L.ontovae_bf16_workspace_bytes(C.byref(self.plan.dims), self.max_batch)
            self._bf16_ws = torch.zeros(nb16 + 256, dtype=torch.uint8, device=self.device)
            self.ensure_grad_buffer()          # the bf16 dW0 product is staged in the gradient buffer
            self._ctx = None
        if self._ctx is None:
            c = _lib.Ctx()
            c.dims = self.plan.dims
            c.lay = self.plan.layout
            c.P, c.M, c.V, c.Gr = _ptr(self.P), _ptr(self.M), _ptr(self.V), _ptr(self.Gr)
            c.bn_rmean, c.bn_rvar, c.bn_nbt = _ptr(self.rmean), _ptr(self.rvar), _ptr(self.nbt)
            i = self._idx
            c.rowptr, c.colidx = _ptr(i["rowptr"]), _ptr(i["colidx"])
            c.cscptr, c.csc_row, c.csc_k = _ptr(i["cscptr"]), _ptr(i["csc_row"]), _ptr(i["csc_k"])
            c.csc_split = _ptr(i["csc_split"])
            c.row_k = _ptr(i["row_k"])
            c.gene_tiles, c.panel_tiles, c.term_tiles = _ptr(i["gene_tiles"]), _ptr(i["panel_tiles"]), _ptr(i["term_tiles"])
            c.tpanel_tiles = _ptr(i["tpanel_tiles"])
            c.tpanel_max_entries = self.plan.tpanel_max_entries
            c.level_begin = self.plan.level_begin.ctypes.data_as(_lib.c_i32p)
            c.level_end = self.plan.level_end.ctypes.data_as(_lib.c_i32p)
            for name, sd, dev in (("sell_fwd", self.plan.sell[0], self._sell[0]), ("sell_bwd", self.plan.sell[1], self._sell[1])):
                t = getattr(c, name)
                t.n_levels, t.n_slices, t.n_chunks = sd["n_levels"], sd["n_slices"], sd["n_chunks"]
                t.all_single_chunk = sd["all_single_chunk"]
                t.level_first = sd["level_first"].ctypes.data_as(_lib.c_i32p)
                t.level_count = sd["level_count"].ctypes.data_as(_lib.c_i32p)
                for f in ("node", "info", "chunk0", "idx", "k", "w", "bias"):
                    setattr(t, f, _ptr(dev[f]))
            c.gsell.n_slices, c.gsell.n_quads = self.plan.gsell["n_slices"], self.plan.gsell["n_quads"]
            for f in ("gene", "quad0", "col", "k", "w"):
                setattr(c.gsell, f, _ptr(self._gsell[f]))
            base = self._ws.data_ptr()
            aligned = (base + 255) // 256 * 256
            c.workspace = C.c_void_p(aligned)
            c.workspace_bytes = self._ws.numel() - (aligned - base)
            c.max_batch = self.max_batch
            c.max_train_batch = self.max_train_batch
            c.drop, c.z_drop, c.seed = self.drop, self.z_drop, self.seed
            c.flags = self.flags
            if self.flags & _lib.FLAG_BF16_GEMM:
                b16 = self._bf16_ws.data_ptr()
                a16 = (b16 + 255) // 256 * 256
                c.bf16_ws = C.c_void_p(a16)
                c.bf16_ws_bytes = self._bf16_ws.numel() - (a16 - b16)
            self._ctx = c
        return self._ctx

    def invalidate(self):
        self._ctx = None

    def set_precision(self, mfma_dtype):
        """'f32' (default: everything in float32, the reference's precision) or 'bf16' (BASELINE.json configs[4]: the two
        large dense products x W0^T and dW0 = dh x on bf16 MFMA with fp32 accumulation; weights and AdamW stay fp32)."""
        if mfma_dtype not in ("f32", "bf16"):
            raise ValueError("mfma_dtype must be 'f32' or 'bf16'")
        if mfma_dtype == "bf16":
            if self.plan.G % 8:
                raise ValueError(f"the bf16 variant needs a gene count that is a multiple of 8, got {self.plan.G}")
            self.flags |= _lib.FLAG_BF16_GEMM
        else:
            self.flags &= ~_lib.FLAG_BF16_GEMM
        self._ctx = None

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def next_offset(self):
        self.rng_offset += 1
        return self.rng_offset

    # ------------------------------------------------------------------ argument checks
    def _x(self, x, xidx):
        p = self.plan
        if xidx is None:
            if x.dim() != 2 or x.shape[1] != p.G:
                raise ValueError(f"expected x of shape [B, {p.G}], got {tuple(x.shape)}")
            x = x.to(device=self.device, dtype=torch.float32).contiguous()
            return x, None, x.shape[0]
        if x.dim() != 2 or x.shape[1] != p.G or x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.device:
            raise ValueError("dataset must be a contiguous float32 [N, G] tensor on the model's device")
        xidx = xidx.to(device=self.device, dtype=torch.int32).contiguous()
        return x, xidx, xidx.shape[0]

    def _eps(self, eps, B):
        if eps is None:
            return None
        eps = torch.as_tensor(eps, dtype=torch.float32, device=self.device).contiguous()
        if tuple(eps.shape) != (B, self.plan.L):
            raise ValueError(f"eps must have shape {(B, self.plan.L)}, got {tuple(eps.shape)}")
        return eps

    # ------------------------------------------------------------------ entry points
    @_on_device
    def forward(self, x, xidx=None, eps=None, train=False, want_recon=True):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        p = self.plan
        recon = torch.empty(B, p.G, dtype=torch.float32, device=self.device) if want_recon else None
        mu = torch.empty(B, p.L, dtype=torch.float32, device=self.device)
        lv = torch.empty(B, p.L, dtype=torch.float32, device=self.device)
        ctx = self.ctx(B, train=bool(train))
        _lib.check(_lib.lib().ontovae_forward(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(),
                                              1 if train else 0, _ptr(recon), _ptr(mu), _ptr(lv), self.stream()), "forward")
        return recon, mu, lv

    @staticmethod
    def adamw_struct(lr, step, betas=ADAM_DEFAULTS["betas"], eps=ADAM_DEFAULTS["eps"], weight_decay=ADAM_DEFAULTS["weight_decay"]):
        b1, b2 = betas
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        a = _lib.AdamW()
        a.decay_mul = 1.0 - lr * weight_decay
        a.one_m_beta1 = 1.0 - b1
        a.beta2 = b2
        a.one_m_beta2 = 1.0 - b2
        a.step_size = lr / bc1
        a.bc2_sqrt = math.sqrt(bc2)
        a.eps = eps
        return a

    @_on_device
    def train_step(self, x, xidx, eps, kl_coeff, adam, loss_out, apply_update=True):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        if apply_update:
            self.ensure_opt_state()
        else:
            self.ensure_grad_buffer()
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_train_step(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(),
                                                 float(kl_coeff), C.byref(adam) if adam is not None else None,
                                                 1 if apply_update else 0, _ptr(loss_out), self.stream()), "train_step")

    @_on_device
    def train_step_dp(self, phase, x, xidx, B_global, eps, kl_coeff, bn_stats, bn_bwd_stats, loss_out, offset):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        self.ensure_grad_buffer()
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_train_step_dp(C.byref(ctx), phase, _ptr(x), _ptr(xidx), B, int(B_global), _ptr(eps),
                                                    offset, float(kl_coeff), _ptr(bn_stats), _ptr(bn_bwd_stats),
                                                    _ptr(loss_out), self.stream()), f"train_step_dp phase {phase}")

    @_on_device
    def adamw_apply(self, adam):
        self.ensure_opt_state()
        self.ensure_grad_buffer()
        ctx = self.ctx(max(self.max_batch, 1))
        _lib.check(_lib.lib().ontovae_adamw_apply(C.byref(ctx), C.byref(adam), self.stream()), "adamw_apply")

    @_on_device
    def adamw_apply_range(self, adam, begin, end):
        self.ensure_opt_state()
        self.ensure_grad_buffer()
        ctx = self.ctx(max(self.max_batch, 1))
        _lib.check(_lib.lib().ontovae_adamw_apply_range(C.byref(ctx), C.byref(adam), int(begin), int(end), self.stream()),
                   "adamw_apply_range")

    @_on_device
    def dp_bn_bwd(self, B, B_global, bn_bwd_stats, offset, send, blm):
        self.ensure_grad_buffer()
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_dp_bn_bwd(C.byref(ctx), int(B), int(B_global), _ptr(bn_bwd_stats), offset, _ptr(send),
                                                int(blm), self.stream()), "dp_bn_bwd")

    @_on_device
    def dw0_rows(self, x, xidx, l0, l1):
        """Rows [l0, l1) of dW0 = dh x of this rank's rows into Gr (after train_step_dp phase 3)."""
        x, xidx, B = self._x(x, xidx)
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_dw0_rows(C.byref(ctx), _ptr(x), _ptr(xidx), B, int(l0), int(l1), self.stream()), "dw0_rows")

    @_on_device
    def dp_pack_dy(self, B, send, blm):
        """After DP phase 1: this rank's block [dy | b0 slot | xhat | local sums] for the single all-gather."""
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_dp_pack_dy(C.byref(ctx), int(B), _ptr(send), int(blm), self.stream()), "dp_pack_dy")

    @_on_device
    def dp_dh_global(self, B_global, sizes, blm, gathered, rstride):
        """Global BatchNorm-backward sums and dh of every rank's columns from the all-gathered blocks, in place."""
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        ctx = self.ctx(max(self.max_batch, 1))
        _lib.check(_lib.lib().ontovae_dp_dh_global(C.byref(ctx), int(B_global), len(sizes), sizes.ctypes.data_as(_lib.c_i32p), int(blm),
                                                   _ptr(gathered), int(rstride), self.stream()), "dp_dh_global")

    @_on_device
    def dw0_global(self, x, xidx, colmap, n_ranks, blm, gathered, rstride, adam):
        """dW0 of the global batch from the all-gathered dh blocks (+ the b0 gradient: sum of the blocks' b0 parts);
        adam=None stores the gradient in Gr."""
        self.ensure_grad_buffer()
        if adam is not None:
            self.ensure_opt_state()
        ctx = self.ctx(max(self.max_batch, 1))
        _lib.check(_lib.lib().ontovae_dw0_global(C.byref(ctx), _ptr(x), _ptr(xidx), _ptr(colmap), int(n_ranks), int(blm), _ptr(gathered),
                                                 int(rstride), C.byref(adam) if adam is not None else None,
                                                 1 if adam is not None else 0, self.stream()), "dw0_global")

    @_on_device
    def eval_step(self, x, xidx, eps, kl_coeff, loss_out):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_eval_step(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(),
                                                float(kl_coeff), _ptr(loss_out), self.stream()), "eval_step")

    @_on_device
    def pass_data(self, x, xidx, eps1, eps2, act_out=None, rec_out=None):
        x, xidx, B = self._x(x, xidx)
        eps1, eps2 = self._eps(eps1, B), self._eps(eps2, B)
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_pass_data(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps1), _ptr(eps2),
                                                self.next_offset(), _ptr(act_out), act_out.stride(0) if act_out is not None else 0,
                                                _ptr(rec_out), rec_out.stride(0) if rec_out is not None else 0, self.stream()),
                   "pass_data")

    @_on_device
    def encode_pre(self, x, xidx, hpre, col0=0):
        """x W0^T of one batch (no bias) into columns [col0, col0 + B) of the feature-major cache hpre [L, ld]."""
        x, xidx, B = self._x(x, xidx)
        if hpre.dtype != torch.float32 or hpre.dim() != 2 or hpre.shape[0] != self.plan.L or hpre.stride(1) != 1 or col0 + B > hpre.shape[1]:
            raise ValueError(f"hpre must be float32 [{self.plan.L}, >= {col0 + B}]")
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_encode_pre(C.byref(ctx), _ptr(x), _ptr(xidx), B, C.c_void_p(hpre.data_ptr() + 4 * col0),
                                                 hpre.stride(0), self.stream()), "encode_pre")

    @_on_device
    def pass_data_perturbed(self, x, xidx, hpre, col0, genes, values, eps1, eps2, act_out=None, rec_out=None, act_fm=None, act_fm_col0=0):
        """pass_data of the batch with columns `genes` of x replaced by `values`, from the cache encode_pre filled.
        act_fm: feature-major [T, ld] matrix whose columns [act_fm_col0, act_fm_col0 + B) receive the activities
        (instead of the sample-major act_out)."""
        x, xidx, B = self._x(x, xidx)
        eps1, eps2 = self._eps(eps1, B), self._eps(eps2, B)
        genes = np.ascontiguousarray(genes, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        if genes.ndim != 1 or genes.shape != values.shape:
            raise ValueError("genes and values must be 1-D and of equal length")
        ctx = self.ctx(B)
        if act_fm is not None:
            if act_out is not None:
                raise ValueError("give act_out or act_fm, not both")
            if act_fm.dtype != torch.float32 or act_fm.dim() != 2 or act_fm.shape[0] != self.plan.T or act_fm.stride(1) != 1 or \
                    act_fm_col0 < 0 or act_fm_col0 + B > act_fm.shape[1]:
                raise ValueError(f"act_fm must be float32 [{self.plan.T}, >= {act_fm_col0 + B}]")
            aptr, ald, afm = C.c_void_p(act_fm.data_ptr() + 4 * act_fm_col0), act_fm.stride(0), 1
        else:
            aptr, ald, afm = _ptr(act_out), act_out.stride(0) if act_out is not None else 0, 0
        _lib.check(_lib.lib().ontovae_pass_data_perturbed(
            C.byref(ctx), _ptr(x), _ptr(xidx), B, C.c_void_p(hpre.data_ptr() + 4 * col0), hpre.stride(0), len(genes),
            genes.ctypes.data_as(_lib.c_i32p), values.ctypes.data_as(_lib.c_f32p), _ptr(eps1), _ptr(eps2), self.next_offset(),
            aptr, ald, afm, _ptr(rec_out),
            rec_out.stride(0) if rec_out is not None else 0, self.stream()), "pass_data_perturbed")

    @_on_device
    def decoder_forward(self, z, act_out=None, rec_out=None, train=False):
        """train: a training pass of a model whose ontology sits in the encoder (ontovae_decoder_forward_train: Dropout(self.drop)
        on the term layers' outputs); decoder_backward on the same batch follows and reuses the pass's Philox offset."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device).contiguous()
        if z.dim() != 2 or z.shape[1] != self.plan.L:
            raise ValueError(f"expected z of shape [B, {self.plan.L}]")
        B = z.shape[0]
        ctx = self.ctx(B, train=train)
        if train:
            if rec_out is None or act_out is not None:
                raise ValueError("a training pass of the operator returns rec_out only")
            self._train_offset = self.next_offset()
            _lib.check(_lib.lib().ontovae_decoder_forward_train(C.byref(ctx), _ptr(z), B, self._train_offset, _ptr(rec_out),
                                                                rec_out.stride(0), self.stream()), "decoder_forward_train")
            return
        _lib.check(_lib.lib().ontovae_decoder_forward(C.byref(ctx), _ptr(z), B, _ptr(act_out),
                                                      act_out.stride(0) if act_out is not None else 0, _ptr(rec_out),
                                                      rec_out.stride(0) if rec_out is not None else 0, self.stream()),
                   "decoder_forward")

    @_on_device
    def decoder_backward(self, drec, adam, apply_update=True, dz_out=None):
        """ontovae_decoder_backward: gradient of the masked operator for the batch decoder_forward(train=True) just ran, AdamW +
        clamp(0) in place (or gradients into Gr), optionally dL/dz."""
        drec = torch.as_tensor(drec, dtype=torch.float32, device=self.device).contiguous()
        if drec.dim() != 2 or drec.shape[1] != self.plan.G:
            raise ValueError(f"expected drec of shape [B, {self.plan.G}]")
        B = drec.shape[0]
        if apply_update:
            self.ensure_opt_state()
        else:
            self.ensure_grad_buffer()
        ctx = self.ctx(B, train=True)
        _lib.check(_lib.lib().ontovae_decoder_backward(C.byref(ctx), _ptr(drec), B, int(getattr(self, "_train_offset", 0)),
                                                       C.byref(adam) if adam is not None else None, 1 if apply_update else 0, _ptr(dz_out),
                                                       dz_out.stride(0) if dz_out is not None else 0, self.stream()), "decoder_backward")
