"""OntoEncVAE on MI355X: class surface of reference onto_vae/vae_model.py:468-905 (ontology-structured ENCODER, dense
Decoder) and of onto_vae/modules.py:161-238 (OntoEncoder), over the C ABI.

The reference class cannot be constructed as it stands: OntoEncoder.__init__ reads `drop` and `z_drop`, which are neither
parameters nor attributes there (modules.py:190-191, a NameError), and it takes the number of top-level terms as its latent
width (modules.py:189) while OntoEncVAE hands the dense Decoder `that * neuronnum` (vae_model.py:501), so the two halves only
fit for neuronnum = 1.  This class is the drop-in for that one working configuration: neuronnum = 1, with drop / z_drop taken
from the constructor (the fixtures come from the unmodified reference with the two names injected as module globals,
tools/gen_golden_encvae.py).

How it runs.  OntoEncoder is the same sparse operator as OntoDecoder, fed with the genes: masked Linear layers whose output is
prepended to their input (modules.py:231-233), then two masked Linear heads over the whole concatenation (:235-236).  The
encoder masks go through the same packing as the decoder masks (MaskPlan; its "latent" input is the gene vector and its last
layer the 2 Z head rows [mu; logvar]), and a step is three library calls:
    ontovae_decoder_forward  (masked encoder: genes -> heads [B, 2 Z])
    ontovae_vae_decode_step  (reparameterisation, dense Decoder, ELBO, backward to the heads, AdamW of the decoder)
    ontovae_decoder_backward (backward of the masked encoder, masked AdamW, clamp of its weights at zero)
Dropout: after the encoder's Linear layers (modules.py:222) the per-level kernels of the operator multiply a layer's output by
Philox keep-scales in training passes (ontovae_decoder_forward_train) and the finished gradient rows by the same scales in
ontovae_decoder_backward; the heads' Dropout(z_drop) and the Decoder's Dropout(drop) are those of the plain VAE path.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .dense_vae import Decoder, DenseEngine, tqdm
from .engine import Engine, norm_device
from .fast_data_loader import FastTensorDataLoader
from .modules import PackedLinear, _LazyMasks, _bind
from .packing import MaskPlan
from .vae_model import ADAM_DEFAULTS, OntoVAE


def _operator_layout(plan):
    """The masked operator alone needs [dec_vals | dec_bias] of the flat parameter buffer; the dense encoder segments that
    ontovae_param_layout lays down for OntoVAE (W0, the 2 L x L heads ...) would be sized by the gene count here."""
    lay = _lib.Layout()
    up = lambda v: (int(v) + 63) // 64 * 64
    lay.dec_vals = 0
    lay.dec_bias = up(max(plan.nnz, 1))
    lay.total = lay.dec_bias + up(plan.C + plan.G)
    return lay


class _HeadLinear(nn.Module):
    """encoder.mu[0] / encoder.logvar[0] (modules.py:200-208, 215-216): rows [row0, row0 + Z) of the operator's last layer.
    state_dict shows the reference's dense `weight` [Z, in] and `bias` [Z]."""

    def __init__(self, engine, row0, rows):
        super().__init__()
        object.__setattr__(self, "_engine", engine)
        self.index = engine.plan.K - 1
        self.row0, self.out_features = int(row0), int(rows)
        self.in_features = engine.plan.layer_shapes[self.index][1]
        self.rebind()

    def rebind(self):
        b = self._engine.layer_bias(self.index)
        _bind(self, "bias", b[self.row0:self.row0 + self.out_features])

    @property
    def weight(self):
        return self._engine.layer_dense(self.index)[self.row0:self.row0 + self.out_features]

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.weight.clone()
        destination[prefix + "bias"] = self.bias.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in ("weight", "bias"):
            if prefix + k not in state_dict and strict:
                missing_keys.append(prefix + k)
        if prefix + "weight" in state_dict:
            w = torch.as_tensor(state_dict[prefix + "weight"], dtype=torch.float32, device=self._engine.device)
            if tuple(w.shape) != (self.out_features, self.in_features):
                error_msgs.append(f"size mismatch for {prefix}weight: {tuple(w.shape)} vs {(self.out_features, self.in_features)}")
            else:
                full = self._engine.layer_dense(self.index)
                full[self.row0:self.row0 + self.out_features] = w
                off = self._engine.set_layer_dense(self.index, full)
                if off != 0.0:
                    import warnings
                    warnings.warn(f"{prefix}weight holds non-zero values (max |w| = {off:g}) outside the ontology mask; they are dropped")
        if prefix + "bias" in state_dict:
            b = torch.as_tensor(state_dict[prefix + "bias"])
            if tuple(b.shape) != tuple(self.bias.shape):
                error_msgs.append(f"size mismatch for {prefix}bias: {tuple(b.shape)} vs {tuple(self.bias.shape)}")
            else:
                self.bias.data.copy_(b)


class OntoEncoder(nn.Module):
    """Ontology-structured encoder (reference modules.py:161-238) for neuronnum = 1: packed masked operator, see the module
    docstring.  Same attributes as the reference: in_features, layer_dims, layer_shapes, masks, latent_dim, drop, z_drop,
    encoder (ModuleList of [Linear, Dropout]), mu, logvar ([Linear, Dropout])."""

    def __init__(self, in_features, layer_dims, mask_list, latent_dim, neuronnum=1, drop=0.2, z_drop=0.5, device=None):
        super().__init__()
        if int(neuronnum) != 1:
            raise ValueError("OntoEncoder's layer widths only fit the rest of OntoEncVAE for neuronnum = 1 (modules.py:189 against "
                             "vae_model.py:501); other values cannot run in the reference either")
        self.device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
        self.in_features = in_features
        self.layer_dims = np.asarray(layer_dims)
        self.layer_shapes = [(np.sum(self.layer_dims[:i + 1]), self.layer_dims[i + 1]) for i in range(len(self.layer_dims) - 1)]
        self.latent_dim = int(self.layer_dims[-1])                      # modules.py:189: the argument is overridden
        self.drop, self.z_drop = drop, z_drop
        ms = [np.asarray(m.detach().cpu().numpy() if hasattr(m, "detach") else m) for m in mask_list]
        if [tuple(m.shape) for m in ms] != [(int(o), int(i)) for i, o in self.layer_shapes]:
            raise ValueError("encoder masks do not have the shapes [(out_i, sum of the widths before it)] of modules.py:185")
        # the operator: every encoder block, then the two heads as ONE last layer of 2 Z rows [mu; logvar] over the same mask
        plan = MaskPlan(ms[:-1] + [np.vstack([ms[-1], ms[-1]])], 1)
        plan.layout = _operator_layout(plan)
        object.__setattr__(self, "_engine", Engine(plan, self.device, float(drop), 0.0))      # drop: Dropout behind every block
        e, Z = self._engine, self.latent_dim
        self.masks = _LazyMasks(e)                                      # (the last one shows the head mask twice)
        self.encoder = nn.ModuleList([nn.Sequential(PackedLinear(e, i), nn.Dropout(p=drop)) for i in range(plan.K - 1)])
        self.mu = nn.Sequential(_HeadLinear(e, 0, Z), nn.Dropout(p=z_drop))
        self.logvar = nn.Sequential(_HeadLinear(e, Z, Z), nn.Dropout(p=z_drop))
        # nn.Linear's default init restricted to the mask (modules.py:211-216: no clamp at construction)
        with torch.no_grad():
            for i in range(plan.K):
                cout, cin = plan.layer_shapes[i]
                bound = 1.0 / np.sqrt(cin)
                v = e.layer_vals(i)
                v.copy_((torch.rand(v.shape) * 2 - 1) * bound)
                b = e.layer_bias(i)
                b.copy_((torch.rand(b.shape) * 2 - 1) * bound)

    def rebind(self):
        for seq in self.encoder:
            seq[0].rebind()
        self.mu[0].rebind()
        self.logvar[0].rebind()

    def heads(self, x, train=False):
        """[B, 2 Z] = [mu | logvar] of the two Linear heads (before their Dropout)."""
        e = self._engine
        x = torch.as_tensor(x, dtype=torch.float32, device=e.device).contiguous()
        h = torch.empty(x.shape[0], 2 * self.latent_dim, dtype=torch.float32, device=e.device)
        e.decoder_forward(x, rec_out=h, train=train)
        return h

    def forward(self, x):
        """modules.py:226-238.  In train mode the blocks' Dropout is applied (Philox masks); the heads' Dropout(z_drop) is part
        of what follows the encoder in OntoEncVAE's passes and refused here."""
        if self.training and self.z_drop:
            raise NotImplementedError("OntoEncoder.forward in train mode with z_drop > 0 runs inside OntoEncVAE.forward / train_round")
        h = self.heads(x, train=self.training)
        Z = self.latent_dim
        return h[:, :Z].contiguous(), h[:, Z:].contiguous()


class _EncVaeAdamW:
    """torch.optim.AdamW(model.parameters(), lr) of reference vae_model.py:698: one step counter and one set of
    hyper-parameters for the two flat buffers (masked encoder, dense decoder)."""

    def __init__(self, model, lr=1e-3, betas=ADAM_DEFAULTS["betas"], eps=ADAM_DEFAULTS["eps"], weight_decay=ADAM_DEFAULTS["weight_decay"]):
        self.model = model
        model.encoder._engine.ensure_opt_state()
        model._dense.ensure_opt_state()
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.step_count = 0

    def next_hyper(self):
        self.step_count += 1
        g = self.param_groups[0]
        return Engine.adamw_struct(g["lr"], self.step_count, g["betas"], g["eps"], g["weight_decay"])

    def zero_grad(self, set_to_none=True):
        pass

    def state_dict(self):
        enc, dec = self.model.encoder._engine, self.model._dense
        return {"state": {"step": self.step_count, "enc_exp_avg": enc.M.detach().clone(), "enc_exp_avg_sq": enc.V.detach().clone(),
                          "dec_exp_avg": dec.M.detach().clone(), "dec_exp_avg_sq": dec.V.detach().clone()},
                "param_groups": [dict(g) for g in self.param_groups], "format": "onto_vae_amd.encvae.flat.v1",
                "packed_order_abi": _lib.ABI_VERSION}

    def load_state_dict(self, sd):
        if sd.get("format") != "onto_vae_amd.encvae.flat.v1" or int(sd.get("packed_order_abi", -1)) != _lib.ABI_VERSION:
            raise ValueError(f"optimizer state in format {sd.get('format')!r} / packed order of ABI {sd.get('packed_order_abi')!r} "
                             f"cannot be taken over by this library (ABI {_lib.ABI_VERSION})")
        enc, dec = self.model.encoder._engine, self.model._dense
        st = sd["state"]
        if tuple(st["enc_exp_avg"].shape) != tuple(enc.P.shape) or tuple(st["dec_exp_avg"].shape) != tuple(dec.P.shape):
            raise ValueError("optimizer state belongs to a model of another size")
        enc.ensure_opt_state(); dec.ensure_opt_state()
        enc.M.copy_(st["enc_exp_avg"]); enc.V.copy_(st["enc_exp_avg_sq"])
        dec.M.copy_(st["dec_exp_avg"]); dec.V.copy_(st["dec_exp_avg_sq"])
        self.step_count = int(st["step"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]


def _as_encvae_opt(model, optimizer, lr):
    if isinstance(optimizer, _EncVaeAdamW):
        return optimizer
    if optimizer is None:
        return _EncVaeAdamW(model, lr=lr)
    if isinstance(optimizer, torch.optim.AdamW):          # the reference's call shape: its live hyper-parameters are adopted
        g = optimizer.param_groups[0]
        if len(optimizer.param_groups) != 1 or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("the fused AdamW takes one parameter group without amsgrad / maximize")
        f = getattr(optimizer, "_ontovae_fused", None)
        if f is None:
            f = optimizer._ontovae_fused = _EncVaeAdamW(model, lr=g["lr"])
        f.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
        return f
    raise TypeError("optimizer must come from OntoEncVAE.train_model (or be a torch.optim.AdamW whose settings are adopted)")


class OntoEncVAE(nn.Module):
    """An ontology-structured encoder combined with a dense decoder (reference vae_model.py:468-525).

    Parameters are those of the reference: ontobj (anything exposing .description, .genes, .masks[key]['encoder'], .data and,
    for terms=, .annot), dataset, top_thresh, bottom_thresh, neuronnum (1, see the module docstring), drop, z_drop."""

    def __init__(self, ontobj, dataset, top_thresh=1000, bottom_thresh=30, neuronnum=1, drop=0.2, z_drop=0.5):
        super().__init__()
        key = str(top_thresh) + '_' + str(bottom_thresh)
        self.ontology = ontobj.description
        self.top = top_thresh
        self.bottom = bottom_thresh
        self.genes = ontobj.genes[key]
        self.in_features = len(self.genes)
        self._mask_src = ontobj.masks[key]['encoder']
        self.layer_dims_enc = np.array([np.asarray(self._mask_src[0]).shape[1]] + [np.asarray(m).shape[0] for m in self._mask_src])
        self.latent_dim = int(self.layer_dims_enc[-1]) * int(neuronnum)
        self.layer_dims_dec = [self.latent_dim]
        self.neuronnum = neuronnum
        self.drop = drop
        self.z_drop = z_drop
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.encoder = OntoEncoder(self.in_features, self.layer_dims_enc, self._mask_src, self.latent_dim, self.neuronnum, drop, z_drop,
                                   device=self.device)
        # the dense Decoder (modules.py:94-154) lives in a dense-VAE engine whose encoder side is a one-unit dummy
        object.__setattr__(self, "_dense", DenseEngine(self.in_features, [1], self.layer_dims_dec, self.latent_dim, self.device, drop, z_drop))
        self.decoder = Decoder(self._dense)
        if dataset not in ontobj.data[key].keys():
            raise ValueError('Available datasets are: ' + ', '.join(list(ontobj.data[key].keys())))
        self.X = ontobj.data[key][dataset]
        self.eps_source = None        # test hook: callable(B) -> eps [B, latent_dim]; None = Philox stream inside the kernels
        self._buf = {}

    @property
    def mask_list(self):
        """vae_model.py:498-499: the encoder masks as float32 tensors (built on access, they are dense)."""
        return [torch.tensor(np.asarray(m), dtype=torch.float32) for m in self._mask_src]

    @property
    def _engine(self):           # what the shared helpers (_batches, loaders) look at: the masked operator's engine
        return self.encoder._engine

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, recurse=True):
        e = self.encoder._engine
        probe = fn(torch.empty(0, dtype=torch.float32, device=e.device))
        if probe.dtype != torch.float32:
            raise TypeError("the HIP OntoEncVAE computes in float32 only")
        if norm_device(probe.device) != e.device:
            e.to(probe.device)
            self._dense.to(probe.device)
            self.device = str(probe.device.type)
            self.encoder.rebind()
            self.decoder.rebind()
            self._buf = {}
        return self

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.encoder._engine.invalidate()
        self._dense.invalidate()

    def _eps(self, B):
        return None if self.eps_source is None else self.eps_source(B)

    def _tmp(self, name, B, width):
        t = self._buf.get(name)
        if t is None or t.shape[0] < B or t.shape[1] != width:
            t = self._buf[name] = torch.empty(max(B, 128), width, dtype=torch.float32, device=self.encoder._engine.device)
        return t[:B]

    def _rows(self, x, xidx):
        """The batch as a contiguous [B, G] matrix on the device: the masked encoder's input (ontovae_gather_rows for an
        HBM-resident dataset and an index list)."""
        e = self.encoder._engine
        if xidx is None:
            return torch.as_tensor(x, dtype=torch.float32, device=e.device).contiguous()
        out = self._tmp("rows", xidx.shape[0], self.in_features)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().ontovae_gather_rows(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(xidx.data_ptr()), xidx.shape[0],
                                                      x.shape[1], C.c_void_p(out.data_ptr()), e.stream()), "gather_rows")
        return out

    # ------------------------------------------------------------------ reference methods
    def reparameterize(self, mu, log_var):
        """vae_model.py:527-538 on caller-supplied tensors (torch ops; the passes below fuse this)."""
        sigma = torch.exp(0.5 * log_var)
        eps = self._eps(mu.shape[0])
        eps = torch.randn_like(sigma) if eps is None else torch.as_tensor(eps, dtype=sigma.dtype, device=sigma.device)
        return mu + eps * sigma

    def get_embedding(self, x):
        """vae_model.py:540-551."""
        h = self.encoder.heads(x, train=self.training)
        return self._dense.decode_forward(h, eps=self._eps(h.shape[0]), train=self.training, want=("z",))["z"]

    def forward(self, x):
        """vae_model.py:553-563.  Returns (reconstruction, mu, log_var), each [B, .] on the device."""
        h = self.encoder.heads(x, train=self.training)
        o = self._dense.decode_forward(h, eps=self._eps(h.shape[0]), train=self.training)
        return o["recon"], o["mu"], o["log_var"]

    def vae_loss(self, reconstruction, mu, log_var, data, kl_coeff, mode='train', run=None):
        """vae_model.py:565-574 on caller-supplied tensors (torch ops; the passes below fuse this)."""
        kl_loss = -0.5 * torch.sum(1. + log_var - mu.pow(2) - log_var.exp())
        rec_loss = torch.sum((reconstruction - data) ** 2)
        if run is not None:
            run["metrics/" + mode + "/kl_loss"].log(kl_loss)
            run["metrics/" + mode + "/rec_loss"].log(rec_loss)
        return torch.mean(rec_loss + kl_coeff * kl_loss)

    _batches = OntoVAE._batches
    _collect = OntoVAE._collect

    def train_step(self, x, xidx, kl_coeff, adam, loss_out, eps=None, apply_update=True):
        """One body of train_round (vae_model.py:598-625) in three library calls; loss_out: device float[3]."""
        rows = self._rows(x, xidx)
        B = rows.shape[0]
        e = self.encoder._engine
        heads = self._tmp("heads", B, 2 * self.latent_dim)
        dheads = self._tmp("dheads", B, 2 * self.latent_dim)
        e.decoder_forward(rows, rec_out=heads, train=True)
        self._dense.decode_step(heads, x if xidx is not None else rows, xidx, eps, kl_coeff, adam, dheads, loss_out, apply_update=apply_update)
        e.decoder_backward(dheads, adam, apply_update=apply_update)

    def train_round(self, dataloader, lr, kl_coeff, optimizer, run=None):
        """vae_model.py:576-629: one epoch of forward, ELBO, backward, masked AdamW, clamp of the encoder weights."""
        self.train()
        opt = _as_encvae_opt(self, optimizer, lr)
        dev = self.encoder._engine.device
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=dev)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            B = x.shape[0] if xidx is None else xidx.shape[0]
            self.train_step(x, xidx, kl_coeff, opt.next_hyper(), losses[n], eps=self._eps(B))
            n += 1
        return self._collect(losses, n, 'train', run) / len(dataloader)

    def val_round(self, dataloader, kl_coeff, run=None):
        """vae_model.py:631-660: eval-mode forward + loss (eps is still sampled)."""
        self.eval()
        dev = self.encoder._engine.device
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=dev)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            rows = self._rows(x, xidx)
            h = self.encoder.heads(rows, train=False)
            self._dense.decode_forward(h, x if xidx is not None else rows, xidx, eps=self._eps(h.shape[0]), train=False, want=(),
                                       kl_coeff=kl_coeff, loss_out=losses[n])
            n += 1
        return self._collect(losses, n, 'val', run) / len(dataloader)

    def train_model(self, modelpath, lr=1e-4, kl_coeff=1e-4, batch_size=128, epochs=300, run=None):
        """vae_model.py:662-720: 80/20 split (seed 42), AdamW defaults, best-val checkpointing."""
        indices = np.random.RandomState(seed=42).permutation(self.X.shape[0])
        X_train_ind = indices[:round(len(indices) * 0.8)]
        X_val_ind = indices[round(len(indices) * 0.8):]
        X_train = torch.tensor(self.X[X_train_ind, :], dtype=torch.float32)
        X_val = torch.tensor(self.X[X_val_ind, :], dtype=torch.float32)
        dev = self.encoder._engine.device
        trainloader = FastTensorDataLoader(X_train, batch_size=batch_size, shuffle=True, device=dev)
        valloader = FastTensorDataLoader(X_val, batch_size=batch_size, shuffle=False, device=dev)

        val_loss_min = float('inf')
        optimizer = _EncVaeAdamW(self, lr=lr)

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

    # ------------------------------------------------------------------ retrieval (vae_model.py:722-905)
    def _pass_data(self, data, output):
        """vae_model.py:722-776: 'act' -> hstack(z, activities of the encoder layers, last layer first); 'rec' -> reconstruction.
        Two eps draws as in the reference (get_embedding, then forward)."""
        self.eval()
        e = self.encoder._engine
        x = torch.as_tensor(data, dtype=torch.float32, device=e.device).contiguous()
        N = x.shape[0]
        p = e.plan
        nterm = p.C - p.L                                     # panel rows above the gene input: the encoder layers' outputs
        out = np.empty((N, self.latent_dim + nterm if output == 'act' else self.in_features), dtype=np.float32)
        step = max(int(getattr(self, "inference_batch", 4096)), 1)
        for b0 in range(0, N, step):
            xb = x[b0:b0 + step]
            B = xb.shape[0]
            heads = torch.empty(B, 2 * self.latent_dim, dtype=torch.float32, device=e.device)
            act = torch.empty(B, p.T, dtype=torch.float32, device=e.device) if output == 'act' else None
            e.decoder_forward(xb, act_out=act, rec_out=heads)
            z = self._dense.decode_forward(heads, eps=self._eps(B), train=False, want=("z",))["z"]
            if output == 'act':
                # activities of ontovae_decoder_forward: [input level | level 1 | ... ]; the reference wants the layers' outputs
                # with the last layer first (vae_model.py:765) behind z
                cols = []
                for lvl in range(p.K - 1, 0, -1):
                    n0, no = p.layer_nodes[lvl - 1]
                    t0 = p.L + sum(p.layer_nodes[j][1] for j in range(lvl - 1))
                    cols.append(act[:, t0:t0 + no])
                out[b0:b0 + B] = torch.cat([z] + cols, dim=1).cpu().numpy()
            else:
                o = self._dense.decode_forward(heads, eps=self._eps(B), train=False, want=("recon",))
                out[b0:b0 + B] = o["recon"].cpu().numpy()
        return out

    def get_pathway_activities(self, ontobj, dataset, terms=None):
        """vae_model.py:779-810."""
        if self.ontology != ontobj.description:
            raise ValueError('Wrong ontology provided, should be ' + self.ontology)
        data = ontobj.data[str(self.top) + '_' + str(self.bottom)][dataset].copy()
        act = self._pass_data(data, 'act')
        if terms is not None:
            annot = ontobj.annot[str(self.top) + '_' + str(self.bottom)]
            term_ind = annot[annot.ID.isin(terms)].index.to_numpy()
            act = act[:, term_ind]
        return act

    def get_reconstructed_values(self, ontobj, dataset, rec_genes=None):
        """vae_model.py:813-844."""
        if self.ontology != ontobj.description:
            raise ValueError('Wrong ontology provided, should be ' + self.ontology)
        data = ontobj.data[str(self.top) + '_' + str(self.bottom)][dataset].copy()
        rec = self._pass_data(data, 'rec')
        if rec_genes is not None:
            onto_genes = ontobj.genes[str(self.top) + '_' + str(self.bottom)]
            gene_ind = np.array([onto_genes.index(g) for g in rec_genes])
            rec = rec[:, gene_ind]
        return rec

    def perturbation(self, ontobj, dataset, genes, values, output='terms', terms=None, rec_genes=None):
        """vae_model.py:847-905."""
        if self.ontology != ontobj.description:
            raise ValueError('Wrong ontology provided, should be ' + self.ontology)
        data = ontobj.data[str(self.top) + '_' + str(self.bottom)][dataset].copy()
        indices = [self.genes.index(g) for g in genes]
        for i in range(len(genes)):
            data[:, indices[i]] = values[i]
        if output == 'terms':
            res = self._pass_data(data, 'act')
        if output == 'genes':
            res = self._pass_data(data, 'rec')
        if terms is not None:
            annot = ontobj.annot[str(self.top) + '_' + str(self.bottom)]
            term_ind = annot[annot.ID.isin(terms)].index.to_numpy()
            res = res[:, term_ind]
        if rec_genes is not None:
            onto_genes = ontobj.genes[str(self.top) + '_' + str(self.bottom)]
            gene_ind = np.array([onto_genes.index(g) for g in rec_genes])
            res = res[:, gene_ind]
        return res
