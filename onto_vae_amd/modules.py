"""Encoder and OntoDecoder with the reference's names, attributes and state_dict layout
(reference onto_vae/modules.py:15-84 and :247-313), backed by the HIP engine.

Both classes are shells: their parameters are views into the engine's flat HBM buffer and their
forward() calls the C ABI.  Nothing here computes on the CPU.
"""
import numpy as np
import torch
import torch.nn as nn

from .engine import Engine
from .packing import MaskPlan


def _bind(module, name, view):
    """(Re)point parameter `name` of `module` at a view of the flat buffer."""
    p = getattr(module, name, None)
    if isinstance(p, nn.Parameter):
        p.data = view
    else:
        setattr(module, name, nn.Parameter(view, requires_grad=False))


class Encoder(nn.Module):
    """Linear -> BatchNorm1d -> Dropout -> ReLU -> (mu, logvar) heads, each head followed by
    Dropout(z_drop).  Same constructor order as the reference (modules.py:33):
    (in_features, latent_dim, layer_dims, drop, z_drop).  Only the configuration OntoVAE uses,
    layer_dims == [latent_dim] (vae_model.py:55, 62-66), is on the hot path."""

    def __init__(self, in_features, latent_dim, layer_dims=[512], drop=0, z_drop=0.5, engine=None):
        super().__init__()
        layer_dims = [int(v) for v in layer_dims]
        latent_dim = int(latent_dim)
        if len(layer_dims) != 1 or layer_dims[0] != latent_dim:
            raise NotImplementedError("the HIP encoder implements the OntoVAE configuration layer_dims == [latent_dim]")
        self.in_features = int(in_features)
        self.layer_dims = layer_dims
        self.layer_nums = []
        self.latent_dim = latent_dim
        self.drop = drop
        self.z_drop = z_drop
        if engine is None:  # stand-alone encoder: an engine with an empty decoder behind it
            plan = MaskPlan([np.zeros((self.in_features, latent_dim), dtype=np.uint8)], 1)
            engine = Engine(plan, "cuda" if torch.cuda.is_available() else "cpu", drop, z_drop)
            self._own_engine = True
        object.__setattr__(self, "_engine", engine)
        lin = nn.Linear(self.in_features, latent_dim)           # default init, then moved into the flat buffer
        bn = nn.BatchNorm1d(latent_dim)
        mu = nn.Linear(latent_dim, latent_dim)
        lv = nn.Linear(latent_dim, latent_dim)
        self.encoder = nn.ModuleList([nn.Sequential(lin, bn, nn.Dropout(p=drop), nn.ReLU())])
        self.mu = nn.Sequential(mu, nn.Dropout(p=z_drop))
        self.logvar = nn.Sequential(lv, nn.Dropout(p=z_drop))
        e = engine
        with torch.no_grad():
            e.view("W0").copy_(lin.weight); e.view("b0").copy_(lin.bias)
            e.view("gamma").copy_(bn.weight); e.view("beta").copy_(bn.bias)
            e.view("Wmu").copy_(mu.weight); e.view("bmu").copy_(mu.bias)
            e.view("Wlv").copy_(lv.weight); e.view("blv").copy_(lv.bias)
        for m in (lin, bn, mu, lv):
            for p in m.parameters():
                p.requires_grad_(False)
        self.rebind()

    def rebind(self):
        e = self._engine
        lin, bn = self.encoder[0][0], self.encoder[0][1]
        _bind(lin, "weight", e.view("W0")); _bind(lin, "bias", e.view("b0"))
        _bind(bn, "weight", e.view("gamma")); _bind(bn, "bias", e.view("beta"))
        _bind(self.mu[0], "weight", e.view("Wmu")); _bind(self.mu[0], "bias", e.view("bmu"))
        _bind(self.logvar[0], "weight", e.view("Wlv")); _bind(self.logvar[0], "bias", e.view("blv"))
        bn._buffers["running_mean"] = e.rmean
        bn._buffers["running_var"] = e.rvar
        bn._buffers["num_batches_tracked"] = e.nbt

    def forward(self, x):
        _, mu, log_var = self._engine.forward(x, train=self.training, want_recon=False)
        return mu, log_var


class PackedLinear(nn.Module):
    """One ontology layer.  state_dict shows the reference's dense `weight` [out, in] and `bias`;
    storage is the packed slice of the engine's value array."""

    def __init__(self, engine, index):
        super().__init__()
        object.__setattr__(self, "_engine", engine)
        self.index = index
        self.out_features, self.in_features = engine.plan.layer_shapes[index]
        self.rebind()

    def rebind(self):
        _bind(self, "packed_weight", self._engine.layer_vals(self.index))
        _bind(self, "bias", self._engine.layer_bias(self.index))

    @property
    def weight(self):
        return self._engine.layer_dense(self.index)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.weight
        destination[prefix + "bias"] = self.bias.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in ("weight", "bias"):
            if prefix + k not in state_dict:
                (missing_keys if strict else []).append(prefix + k)
        if prefix + "weight" in state_dict:
            try:
                off = self._engine.set_layer_dense(self.index, state_dict[prefix + "weight"])
                if off != 0.0:
                    import warnings
                    warnings.warn(f"{prefix}weight holds non-zero values (max |w| = {off:g}) outside the ontology mask; "
                                  "they are dropped (the reference would keep using them in forward)")
            except ValueError as e:
                error_msgs.append(str(e))
        if prefix + "bias" in state_dict:
            b = torch.as_tensor(state_dict[prefix + "bias"])
            if tuple(b.shape) != tuple(self.bias.shape):
                error_msgs.append(f"size mismatch for {prefix}bias: {tuple(b.shape)} vs {tuple(self.bias.shape)}")
            else:
                self.bias.data.copy_(b)
        for k in list(state_dict.keys()):
            if k.startswith(prefix) and k[len(prefix):] not in ("weight", "bias") and strict:
                unexpected_keys.append(k)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, packed={self.packed_weight.numel()}"


class _LazyMasks:
    """decoder.masks of the reference (modules.py:270-275): expanded dense float32 masks on the
    device, materialised only when somebody indexes them."""

    def __init__(self, engine):
        self._e = engine

    def __len__(self):
        return self._e.plan.K

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return torch.from_numpy(self._e.plan.unpack_mask(i)).to(self._e.device, torch.float32)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class OntoDecoder(nn.Module):
    """Ontology-structured linear decoder (reference modules.py:247-313): K masked, non-negative
    Linear layers, the output of each prepended to its input.  Here: one packed sparse operator
    over the [panel | genes] node space, run level by level."""

    def __init__(self, in_features, layer_dims, mask_list, latent_dim, neuronnum=3, engine=None):
        super().__init__()
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.in_features = in_features
        layer_dims = np.asarray(layer_dims)
        self.layer_dims = np.hstack([layer_dims[:-1] * neuronnum, layer_dims[-1]])
        self.layer_shapes = [(np.sum(self.layer_dims[:i + 1]), self.layer_dims[i + 1]) for i in range(len(self.layer_dims) - 1)]
        self.latent_dim = latent_dim
        if engine is None:
            engine = Engine(MaskPlan(mask_list, neuronnum), self.device)
        object.__setattr__(self, "_engine", engine)
        plan = engine.plan
        assert [(int(a), int(b)) for a, b in self.layer_shapes] == [(cin, cout) for cout, cin in plan.layer_shapes]
        self.masks = _LazyMasks(engine)
        self.decoder = nn.ModuleList([nn.Sequential(PackedLinear(engine, i)) for i in range(plan.K)])
        # nn.Linear's default init restricted to the mask, then clamp(0)  (modules.py:291-296)
        with torch.no_grad():
            for i in range(plan.K):
                cout, cin = plan.layer_shapes[i]
                bound = 1.0 / np.sqrt(cin)
                v = engine.layer_vals(i)
                v.copy_(((torch.rand(v.shape) * 2 - 1) * bound).clamp_(min=0))
                b = engine.layer_bias(i)
                b.copy_((torch.rand(b.shape) * 2 - 1) * bound)

    def rebind(self):
        for seq in self.decoder:
            seq[0].rebind()

    def forward(self, z):
        rec = torch.empty(z.shape[0], self._engine.plan.G, dtype=torch.float32, device=self._engine.device)
        self._engine.decoder_forward(z, rec_out=rec)
        return rec
