"""The plain VAE sibling on MI355X: class surface of reference onto_vae/vae_model.py:913-1140 (VAE) over
onto_vae/modules.py:15-84 (Encoder, any number of blocks) and :94-154 (Decoder), with the arithmetic in libontovae_hip.so
(ontovae_vae_* of include/ontovae_hip.h).  Constructor, method names, arguments, printed strings and the state_dict keys
follow the reference; as for OntoVAE the modules are shells whose parameters are views into one flat HBM buffer.
Single GPU (the data-parallel step exists for OntoVAE only)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import Engine, norm_device, _on_device, _ptr
from .fast_data_loader import FastTensorDataLoader
from .modules import _bind

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **kw):
        return it


class DenseEngine:
    """Device state of one dense VAE: flat parameters / moments / gradients, BatchNorm statistics, workspace, ontovae_vae_ctx_t."""

    def __init__(self, in_features, layer_dims_enc, layer_dims_dec, latent_dim, device, drop=0.0, z_drop=0.0):
        enc, dec = [int(v) for v in layer_dims_enc], [int(v) for v in layer_dims_dec]
        if not 1 <= len(enc) <= _lib.VAE_MAXL or not 1 <= len(dec) <= _lib.VAE_MAXL:
            raise ValueError(f"layer_dims_enc / layer_dims_dec need between 1 and {_lib.VAE_MAXL} entries")
        d = _lib.VaeDims()
        d.G, d.Z, d.n_enc, d.n_dec = int(in_features), int(latent_dim), len(enc), len(dec)
        for i, v in enumerate(enc):
            d.enc[i] = v
        for i, v in enumerate(dec):
            d.dec[i] = v
        self.dims, self.enc, self.dec = d, enc, dec
        self.G, self.Z = int(in_features), int(latent_dim)
        self.layout = _lib.VaeLayout()
        _lib.check(_lib.lib().ontovae_vae_layout(C.byref(d), C.byref(self.layout)), "vae_layout")
        self.device = norm_device(device)
        self.drop, self.z_drop = float(drop), float(z_drop)
        lay = self.layout
        self.P = torch.zeros(lay.total, dtype=torch.float32, device=self.device)
        self.M = self.V = self.Gr = None
        self.bn_state = torch.zeros(lay.bn_total, dtype=torch.float32, device=self.device)
        self.nbt = torch.zeros(len(enc) + len(dec), dtype=torch.int64, device=self.device)
        for side, widths in (("enc", enc), ("dec", dec)):
            for i, w in enumerate(widths):
                self.bn_view(side, i)[1].fill_(1.0)
        self._ws, self.max_batch, self._ctx = None, 0, None
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.rng_offset = 0

    # ------------------------------------------------------------------ views
    def block_shape(self, side, i):
        widths = self.enc if side == "enc" else self.dec
        first = self.G if side == "enc" else self.Z
        return widths[i], (first if i == 0 else widths[i - 1])

    def block_views(self, side, i, buf=None):
        """(W [out, in], b, gamma, beta) of block i of the encoder ('enc') or decoder ('dec')."""
        buf = self.P if buf is None else buf
        out, cin = self.block_shape(side, i)
        lay = self.layout
        o = lambda n: getattr(lay, side + "_" + n)[i]
        return (buf[o("W"):o("W") + out * cin].view(out, cin), buf[o("b"):o("b") + out],
                buf[o("gamma"):o("gamma") + out], buf[o("beta"):o("beta") + out])

    def bn_view(self, side, i):
        out, _ = self.block_shape(side, i)
        o = getattr(self.layout, side + "_bn")[i]
        return self.bn_state[o:o + out], self.bn_state[o + out:o + 2 * out]

    def nbt_view(self, side, i):
        k = i if side == "enc" else len(self.enc) + i
        return self.nbt[k]

    def head_views(self, buf=None):
        """(Wmu, bmu, Wlv, blv)."""
        buf = self.P if buf is None else buf
        Z, H, lay = self.Z, self.enc[-1], self.layout
        return (buf[lay.Wh:lay.Wh + Z * H].view(Z, H), buf[lay.bh:lay.bh + Z],
                buf[lay.Wh + Z * H:lay.Wh + 2 * Z * H].view(Z, H), buf[lay.bh + Z:lay.bh + 2 * Z])

    def out_views(self, buf=None):
        buf = self.P if buf is None else buf
        G, H, lay = self.G, self.dec[-1], self.layout
        return buf[lay.Wout:lay.Wout + G * H].view(G, H), buf[lay.bout:lay.bout + G]

    # ------------------------------------------------------------------ plumbing
    def to(self, device):
        device = norm_device(device)
        if device != self.device:
            self.device = device
            for n in ("P", "M", "V", "Gr", "bn_state", "nbt"):
                v = getattr(self, n)
                if v is not None:
                    setattr(self, n, v.to(device))
            self._ws, self.max_batch, self._ctx = None, 0, None
        return self

    def ensure_opt_state(self):
        if self.M is None:
            self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P)
            self._ctx = None

    def ensure_grad_buffer(self):
        if self.Gr is None:
            self.Gr = torch.zeros_like(self.P)
            self._ctx = None

    def invalidate(self):
        self._ctx = None

    _require_gpu = Engine._require_gpu
    stream = Engine.stream
    next_offset = Engine.next_offset
    adamw_struct = staticmethod(Engine.adamw_struct)

    def ctx(self, batch):
        self._require_gpu()
        if batch > self.max_batch:
            mb = max(int(batch), 128)
            nbytes = _lib.lib().ontovae_vae_workspace_bytes(C.byref(self.dims), mb)
            if nbytes <= 0:
                raise _lib.OntoVAEHipError("vae workspace sizing failed")
            self._ws = None
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self.max_batch, self._ctx = mb, None
        if self._ctx is None:
            c = _lib.VaeCtx()
            c.dims = self.dims
            c.P, c.M, c.V, c.Gr = _ptr(self.P), _ptr(self.M), _ptr(self.V), _ptr(self.Gr)
            c.bn_state, c.bn_nbt = _ptr(self.bn_state), _ptr(self.nbt)
            base = self._ws.data_ptr()
            aligned = (base + 255) // 256 * 256
            c.workspace, c.workspace_bytes = C.c_void_p(aligned), self._ws.numel() - (aligned - base)
            c.max_batch, c.drop, c.z_drop, c.seed = self.max_batch, self.drop, self.z_drop, self.seed
            self._ctx = c
        return self._ctx

    def _x(self, x, xidx):
        if xidx is None:
            if x.dim() != 2 or x.shape[1] != self.G:
                raise ValueError(f"expected x of shape [B, {self.G}], got {tuple(x.shape)}")
            x = x.to(device=self.device, dtype=torch.float32).contiguous()
            return x, None, x.shape[0]
        if x.dim() != 2 or x.shape[1] != self.G or x.dtype != torch.float32 or not x.is_contiguous() or x.device != self.device:
            raise ValueError("dataset must be a contiguous float32 [N, G] tensor on the model's device")
        xidx = xidx.to(device=self.device, dtype=torch.int32).contiguous()
        return x, xidx, xidx.shape[0]

    def _eps(self, eps, B):
        if eps is None:
            return None
        eps = torch.as_tensor(eps, dtype=torch.float32, device=self.device).contiguous()
        if tuple(eps.shape) != (B, self.Z):
            raise ValueError(f"eps must have shape {(B, self.Z)}, got {tuple(eps.shape)}")
        return eps

    # ------------------------------------------------------------------ entry points
    @_on_device
    def forward(self, x, xidx=None, eps=None, train=False, want=("recon", "mu", "log_var"), kl_coeff=0.0, loss_out=None):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        width = dict(recon=self.G, mu=self.Z, log_var=self.Z, z=self.Z)
        out = {k: (torch.empty(B, width[k], dtype=torch.float32, device=self.device) if k in want else None) for k in width}
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_vae_forward(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(), 1 if train else 0,
                                                  float(kl_coeff), _ptr(out["recon"]), _ptr(out["mu"]), _ptr(out["log_var"]), _ptr(out["z"]),
                                                  _ptr(loss_out), self.stream()), "vae_forward")
        return out

    @_on_device
    def train_step(self, x, xidx, eps, kl_coeff, adam, loss_out, apply_update=True):
        x, xidx, B = self._x(x, xidx)
        eps = self._eps(eps, B)
        if apply_update:
            self.ensure_opt_state()
        else:
            self.ensure_grad_buffer()
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_vae_train_step(C.byref(ctx), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(), float(kl_coeff),
                                                     C.byref(adam) if adam is not None else None, 1 if apply_update else 0, _ptr(loss_out),
                                                     self.stream()), "vae_train_step")

    def eval_step(self, x, xidx, eps, kl_coeff, loss_out):
        self.forward(x, xidx, eps, train=False, want=(), kl_coeff=kl_coeff, loss_out=loss_out)

    # ---- OntoEncVAE: the masked encoder's heads come in, their gradient goes out (ontovae_vae_decode_forward / _step)
    def _heads(self, heads):
        if heads.dim() != 2 or heads.shape[1] != 2 * self.Z or heads.dtype != torch.float32 or not heads.is_contiguous() or heads.device != self.device:
            raise ValueError(f"heads must be a contiguous float32 [B, {2 * self.Z}] tensor on the model's device")
        return heads, heads.shape[0]

    @_on_device
    def decode_forward(self, heads, x=None, xidx=None, eps=None, train=False, want=("recon", "mu", "log_var"), kl_coeff=0.0, loss_out=None):
        heads, B = self._heads(heads)
        if x is not None:
            x, xidx, Bx = self._x(x, xidx)
            if Bx != B:
                raise ValueError("heads and x disagree on the batch size")
        eps = self._eps(eps, B)
        width = dict(recon=self.G, mu=self.Z, log_var=self.Z, z=self.Z)
        out = {k: (torch.empty(B, width[k], dtype=torch.float32, device=self.device) if k in want else None) for k in width}
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_vae_decode_forward(C.byref(ctx), _ptr(heads), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(),
                                                         1 if train else 0, float(kl_coeff), _ptr(out["recon"]), _ptr(out["mu"]),
                                                         _ptr(out["log_var"]), _ptr(out["z"]), _ptr(loss_out), self.stream()), "vae_decode_forward")
        return out

    @_on_device
    def decode_step(self, heads, x, xidx, eps, kl_coeff, adam, dheads_out, loss_out, apply_update=True):
        heads, B = self._heads(heads)
        x, xidx, Bx = self._x(x, xidx)
        if Bx != B:
            raise ValueError("heads and x disagree on the batch size")
        eps = self._eps(eps, B)
        if apply_update:
            self.ensure_opt_state()
        else:
            self.ensure_grad_buffer()
        ctx = self.ctx(B)
        _lib.check(_lib.lib().ontovae_vae_decode_step(C.byref(ctx), _ptr(heads), _ptr(x), _ptr(xidx), B, _ptr(eps), self.next_offset(),
                                                      float(kl_coeff), C.byref(adam) if adam is not None else None, 1 if apply_update else 0,
                                                      _ptr(dheads_out), _ptr(loss_out), self.stream()), "vae_decode_step")

    @_on_device
    def adamw_apply(self, adam):
        self.ensure_opt_state()
        self.ensure_grad_buffer()
        ctx = self.ctx(max(self.max_batch, 1))
        _lib.check(_lib.lib().ontovae_vae_adamw_apply(C.byref(ctx), C.byref(adam), self.stream()), "vae_adamw_apply")


def _block(cin, cout, drop):
    return nn.Sequential(nn.Linear(cin, cout), nn.BatchNorm1d(cout), nn.Dropout(p=drop), nn.ReLU())


def _adopt(engine, seq, views, bn=None, nbt=None):
    """Move a freshly initialised [Linear(, BatchNorm1d)] into the flat buffer and point its parameters there."""
    with torch.no_grad():
        views[0].copy_(seq[0].weight); views[1].copy_(seq[0].bias)
        if len(views) == 4:
            views[2].copy_(seq[1].weight); views[3].copy_(seq[1].bias)


def _rebind(seq, views, bn=None, nbt=None):
    _bind(seq[0], "weight", views[0]); _bind(seq[0], "bias", views[1])
    if len(views) == 4:
        _bind(seq[1], "weight", views[2]); _bind(seq[1], "bias", views[3])
        seq[1]._buffers["running_mean"], seq[1]._buffers["running_var"] = bn
        seq[1]._buffers["num_batches_tracked"] = nbt


class DenseEncoder(nn.Module):
    """Reference Encoder (modules.py:15-84) with any layer_dims: blocks, then the mu / logvar heads."""

    def __init__(self, engine):
        super().__init__()
        object.__setattr__(self, "_engine", engine)
        e = engine
        self.in_features, self.layer_dims, self.latent_dim, self.drop, self.z_drop = e.G, list(e.enc), e.Z, e.drop, e.z_drop
        self.layer_nums = [self.layer_dims[i:i + 2] for i in range(len(self.layer_dims) - 1)]
        self.encoder = nn.ModuleList([_block(*reversed(e.block_shape("enc", i)), e.drop) for i in range(len(e.enc))])
        self.mu = nn.Sequential(nn.Linear(e.enc[-1], e.Z), nn.Dropout(p=e.z_drop))
        self.logvar = nn.Sequential(nn.Linear(e.enc[-1], e.Z), nn.Dropout(p=e.z_drop))
        for i, seq in enumerate(self.encoder):
            _adopt(e, seq, e.block_views("enc", i))
        h = e.head_views()
        _adopt(e, self.mu, h[:2]); _adopt(e, self.logvar, h[2:])
        for p in self.parameters():
            p.requires_grad_(False)
        self.rebind()

    def rebind(self):
        e = self._engine
        for i, seq in enumerate(self.encoder):
            _rebind(seq, e.block_views("enc", i), e.bn_view("enc", i), e.nbt_view("enc", i))
        h = e.head_views()
        _rebind(self.mu, h[:2]); _rebind(self.logvar, h[2:])

    def forward(self, x):
        o = self._engine.forward(torch.as_tensor(x), train=self.training, want=("mu", "log_var"))
        return o["mu"], o["log_var"]


class Decoder(nn.Module):
    """Reference Decoder (modules.py:94-154): blocks, then a Linear to the genes.  It runs inside VAE.forward; a call on a
    z of the caller's is not part of the hot path and is refused."""

    def __init__(self, engine):
        super().__init__()
        object.__setattr__(self, "_engine", engine)
        e = engine
        self.in_features, self.layer_dims, self.latent_dim, self.drop = e.G, list(e.dec), e.Z, e.drop
        self.layer_nums = [self.layer_dims[i:i + 2] for i in range(len(self.layer_dims) - 1)]
        self.decoder = nn.ModuleList([_block(*reversed(e.block_shape("dec", i)), e.drop) for i in range(len(e.dec))] +
                                     [nn.Sequential(nn.Linear(e.dec[-1], e.G))])
        for i in range(len(e.dec)):
            _adopt(e, self.decoder[i], e.block_views("dec", i))
        _adopt(e, self.decoder[-1], e.out_views())
        for p in self.parameters():
            p.requires_grad_(False)
        self.rebind()

    def rebind(self):
        e = self._engine
        for i in range(len(e.dec)):
            _rebind(self.decoder[i], e.block_views("dec", i), e.bn_view("dec", i), e.nbt_view("dec", i))
        _rebind(self.decoder[-1], e.out_views())

    def forward(self, z):
        raise NotImplementedError("the dense decoder runs inside VAE.forward (ontovae_vae_forward); there is no entry point for a caller's z")


class VAE(nn.Module):
    """A standard VAE without ontology (reference vae_model.py:913-955).

    Parameters as in the reference: data (2D numpy array, samples in rows), layer_dims_enc, layer_dims_dec, latent_dim,
    drop, z_drop."""

    def __init__(self, data, layer_dims_enc=[1000], layer_dims_dec=[1000], latent_dim=128, drop=0.2, z_drop=0.5):
        super().__init__()
        self.X = data
        self.in_features = self.X.shape[1]
        self.layer_dims_enc = layer_dims_enc
        self.layer_dims_dec = layer_dims_dec
        self.latent_dim = latent_dim
        self.drop = drop
        self.z_drop = z_drop
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        object.__setattr__(self, "_engine", DenseEngine(self.in_features, layer_dims_enc, layer_dims_dec, latent_dim, self.device, drop, z_drop))
        self.encoder = DenseEncoder(self._engine)
        self.decoder = Decoder(self._engine)
        self.eps_source = None        # test hook: callable(B) -> eps [B, latent_dim]; None = Philox stream inside the kernels

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._engine.device))
        if probe.dtype != torch.float32:
            raise TypeError("the HIP VAE computes in float32 only")
        if norm_device(probe.device) != self._engine.device:
            self._engine.to(probe.device)
            self.device = str(probe.device.type)
            self.encoder.rebind()
            self.decoder.rebind()
        return self

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._engine.invalidate()

    def _eps(self, B):
        return None if self.eps_source is None else self.eps_source(B)

    # ------------------------------------------------------------------ reference methods
    def reparameterize(self, mu, log_var):
        """vae_model.py:957-968 on caller-supplied tensors (torch ops; the passes below fuse this)."""
        sigma = torch.exp(0.5 * log_var)
        eps = self._eps(mu.shape[0])
        eps = torch.randn_like(sigma) if eps is None else torch.as_tensor(eps, dtype=sigma.dtype, device=sigma.device)
        return mu + eps * sigma

    def get_embedding(self, x):
        """vae_model.py:970-981."""
        x = torch.as_tensor(x)
        return self._engine.forward(x, eps=self._eps(x.shape[0]), train=self.training, want=("z",))["z"]

    def forward(self, x):
        """vae_model.py:983-993.  Returns (reconstruction, mu, log_var), each [B, .] on the device."""
        x = torch.as_tensor(x)
        o = self._engine.forward(x, eps=self._eps(x.shape[0]), train=self.training)
        return o["recon"], o["mu"], o["log_var"]

    def vae_loss(self, reconstruction, mu, log_var, data, kl_coeff, mode='train', run=None):
        """vae_model.py:995-1004 on caller-supplied tensors (torch ops; the passes below fuse this)."""
        kl_loss = -0.5 * torch.sum(1. + log_var - mu.pow(2) - log_var.exp())
        rec_loss = torch.sum((reconstruction - data) ** 2)
        if run is not None:
            run["metrics/" + mode + "/kl_loss"].log(kl_loss)
            run["metrics/" + mode + "/rec_loss"].log(rec_loss)
        return torch.mean(rec_loss + kl_coeff * kl_loss)

    def _batches(self, dataloader):
        from .vae_model import OntoVAE
        return OntoVAE._batches(self, dataloader)

    def _collect(self, losses, n, mode, run):
        from .vae_model import OntoVAE
        return OntoVAE._collect(self, losses, n, mode, run)

    def train_round(self, dataloader, lr, kl_coeff, optimizer, run=None):
        """vae_model.py:1006-1047: one epoch of forward, ELBO, backward, AdamW."""
        from .vae_model import _as_fused
        self.train()
        opt = _as_fused(self, optimizer, lr)
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=self._engine.device)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            B = x.shape[0] if xidx is None else xidx.shape[0]
            self._engine.train_step(x, xidx, self._eps(B), kl_coeff, opt.next_hyper(), losses[n])
            n += 1
        return self._collect(losses, n, 'train', run) / len(dataloader)

    def val_round(self, dataloader, kl_coeff, run=None):
        """vae_model.py:1049-1080: eval-mode forward + loss (eps is still sampled)."""
        self.eval()
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=self._engine.device)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            B = x.shape[0] if xidx is None else xidx.shape[0]
            self._engine.eval_step(x, xidx, self._eps(B), kl_coeff, losses[n])
            n += 1
        return self._collect(losses, n, 'val', run) / len(dataloader)

    def train_model(self, modelpath, lr=1e-4, kl_coeff=1e-4, batch_size=128, epochs=300, run=None):
        """vae_model.py:1082-1140: 80/20 split (seed 42), AdamW defaults, best-val checkpointing."""
        from .vae_model import FusedAdamW
        indices = np.random.RandomState(seed=42).permutation(self.X.shape[0])
        X_train_ind = indices[:round(len(indices) * 0.8)]
        X_val_ind = indices[round(len(indices) * 0.8):]
        X_train = torch.tensor(self.X[X_train_ind, :], dtype=torch.float32)
        X_val = torch.tensor(self.X[X_val_ind, :], dtype=torch.float32)
        dev = self._engine.device
        trainloader = FastTensorDataLoader(X_train, batch_size=batch_size, shuffle=True, device=dev)
        valloader = FastTensorDataLoader(X_val, batch_size=batch_size, shuffle=False, device=dev)

        val_loss_min = float('inf')
        optimizer = FusedAdamW(self, lr=lr)

        for epoch in range(epochs):
            print(f"Epoch {epoch+1} of {epochs}")
            train_epoch_loss = self.train_round(trainloader, lr, kl_coeff, optimizer, run)
            val_epoch_loss = self.val_round(valloader, kl_coeff, run)

            if run is not None:
                run["metrics/train/loss"].log(train_epoch_loss)
                run["metrics/val/loss"].log(val_epoch_loss)

            if val_epoch_loss < val_loss_min:
                print('New best model!')
                if modelpath is not None:
                    torch.save({
                        'epoch': epoch,
                        'model_state_dict': self.state_dict(),
                        'optimizer_state_dict': optimizer.state_dict(),
                        'loss': val_epoch_loss,
                    }, modelpath)
                val_loss_min = val_epoch_loss

            print(f"Train Loss: {train_epoch_loss:.4f}")
            print(f"Val Loss: {val_epoch_loss:.4f}")
