"""CPU oracle for OntoEncVAE (reference onto_vae/vae_model.py:468-905, onto_vae/modules.py:161-238).  TEST INFRASTRUCTURE ONLY.

Dense numpy restatement with a hand-written backward: the masked encoder (this file) in front of the dense Decoder and the
AdamW / BatchNorm arithmetic of oracle/vae_oracle.py.  Pinned by tests/golden/encvae_*.npz, produced by the imported reference
itself (tools/gen_golden_encvae.py: the unmodified classes with `drop` / `z_drop` injected as module globals, neuronnum = 1 --
the one configuration the reference can run, see onto_vae_amd/enc_vae_model.py).

Reference map:
  OntoEncoder.forward  modules.py:226-238   out = x; per block c = Linear(out) [Dropout]; out = cat(c, out); mu / logvar = Linear(out) [Dropout]
  masks                modules.py:211-216   weights multiplied by the masks at construction (no clamp there)
  train_round body     vae_model.py:598-625 ELBO, backward, gradient masks (:611-616), AdamW (:619), clamp(0) of the encoder weights (:622-625)
  val_round body       vae_model.py:651-655
eps supplied.  Dropout: off unless the caller hands in keep-scales (`scales`: {"encoder.encoder.i": [B, width_i], "mu" /
"logvar": [B, Z], "decoder.decoder.i": [B, width]}, entries 0 or 1 / (1 - p)), e.g. those of the HIP path's Philox streams.
"""
import numpy as np

from vae_oracle import OracleVAE


class OracleOntoEncVAE(OracleVAE):
    def __init__(self, enc_masks, layer_dims_dec=None, dtype=np.float32):
        masks = [np.asarray(m, dtype=dtype) for m in enc_masks]
        G, Z = masks[0].shape[1], masks[-1].shape[0]
        super().__init__(G, (), tuple(layer_dims_dec) if layer_dims_dec is not None else (Z,), Z, dtype)
        self.masks = masks
        self.nl = len(masks) - 1          # masked blocks in front of the heads

    def enc_weight_names(self):
        return [f"encoder.encoder.{i}.0.weight" for i in range(self.nl)] + ["encoder.mu.0.weight", "encoder.logvar.0.weight"]

    def param_names(self):
        names = []
        for i in range(self.nl):
            names += [f"encoder.encoder.{i}.0.weight", f"encoder.encoder.{i}.0.bias"]
        return names + super().param_names()

    # ------------------------------------------------------------------ forward
    def encoder_forward(self, x, scales=None):
        """mu, logvar (before the heads' Dropout), the concatenation the heads see, the inputs of every block and the blocks'
        Linear outputs (what the reference's forward hooks see, vae_model.py:757-766).  scales: keep-scales of the blocks'
        Dropout (modules.py:222), {"encoder.encoder.i": [B, width]}, train mode only."""
        p, dt = self.p, self.dt
        out = np.asarray(x, dtype=dt)
        ins, acts = [], []
        for i in range(self.nl):
            ins.append(out)
            c = out @ p[f"encoder.encoder.{i}.0.weight"].T + p[f"encoder.encoder.{i}.0.bias"]
            acts.append(c)
            if scales is not None and f"encoder.encoder.{i}" in scales:
                c = c * np.asarray(scales[f"encoder.encoder.{i}"], dtype=dt)
            out = np.concatenate([c, out], axis=1)
        mu = out @ p["encoder.mu.0.weight"].T + p["encoder.mu.0.bias"]
        lv = out @ p["encoder.logvar.0.weight"].T + p["encoder.logvar.0.bias"]
        return mu.astype(dt), lv.astype(dt), out, ins, acts

    def forward(self, x, eps, train, cache=None, scales=None):
        p, dt = self.p, self.dt
        cache = [] if cache is None else cache
        sc = (lambda k: None) if scales is None else (lambda k: np.asarray(scales[k], dtype=dt))
        mu, lv, top, ins, acts = self.encoder_forward(x, scales if train else None)
        if scales is not None:
            mu, lv = mu * sc("mu"), lv * sc("logvar")
        z = (mu + np.asarray(eps, dtype=dt) * np.exp(dt(0.5) * lv)).astype(dt)
        a = z
        for i in range(len(self.dec)):
            a = self._block(f"decoder.decoder.{i}", a, train, cache, sc(f"decoder.decoder.{i}"))
        n = len(self.dec)
        recon = a @ p[f"decoder.decoder.{n}.0.weight"].T + p[f"decoder.decoder.{n}.0.bias"]
        return recon.astype(dt), mu.astype(dt), lv.astype(dt), dict(top=top, z=z, last=a, cache=cache, ins=ins, acts=acts)

    def pass_data(self, x, eps1, eps2, output):
        """_pass_data (vae_model.py:722-776): 'act' = hstack(z of the first draw, block outputs with the last block first)."""
        dt = self.dt
        mu, lv, _, _, acts = self.encoder_forward(x)
        z = mu + np.asarray(eps1, dtype=dt) * np.exp(dt(0.5) * lv)
        if output == 'act':
            return np.hstack([z] + acts[::-1]).astype(dt)
        return self.forward(x, eps2, False)[0]

    # ------------------------------------------------------------------ backward
    def gradients(self, x, eps, kl_coeff, scales=None):
        p, dt = self.p, self.dt
        x = np.asarray(x, dtype=dt)
        eps = np.asarray(eps, dtype=dt)
        recon, mu, lv, f = self.forward(x, eps, True, scales=scales)
        rec, kl, loss = self.vae_loss(recon, mu, lv, x, kl_coeff)
        g = {}
        n = len(self.dec)
        d = dt(2) * (recon - x)
        g[f"decoder.decoder.{n}.0.weight"] = d.T @ f["last"]
        g[f"decoder.decoder.{n}.0.bias"] = d.sum(axis=0, dtype=dt)
        d = d @ p[f"decoder.decoder.{n}.0.weight"]
        for c in reversed(f["cache"]):
            d = self._block_bwd(c, d, g)
        kc = dt(kl_coeff)
        sigma = np.exp(dt(0.5) * lv)
        d_mu = d + kc * mu
        d_lv = d * eps * dt(0.5) * sigma + kc * dt(0.5) * (np.exp(lv) - dt(1))
        if scales is not None:
            d_mu, d_lv = d_mu * np.asarray(scales["mu"], dtype=dt), d_lv * np.asarray(scales["logvar"], dtype=dt)
        top = f["top"]
        g["encoder.mu.0.weight"] = (d_mu.T @ top) * self.masks[-1]               # vae_model.py:615-616
        g["encoder.mu.0.bias"] = d_mu.sum(axis=0, dtype=dt)
        g["encoder.logvar.0.weight"] = (d_lv.T @ top) * self.masks[-1]
        g["encoder.logvar.0.bias"] = d_lv.sum(axis=0, dtype=dt)
        d_out = d_mu @ p["encoder.mu.0.weight"] + d_lv @ p["encoder.logvar.0.weight"]
        for i in reversed(range(self.nl)):
            W = p[f"encoder.encoder.{i}.0.weight"]
            no = W.shape[0]
            d_c, d_prev = d_out[:, :no], d_out[:, no:]
            if scales is not None and f"encoder.encoder.{i}" in scales:
                d_c = d_c * np.asarray(scales[f"encoder.encoder.{i}"], dtype=dt)
            g[f"encoder.encoder.{i}.0.weight"] = (d_c.T @ f["ins"][i]) * self.masks[i]      # vae_model.py:611-612
            g[f"encoder.encoder.{i}.0.bias"] = d_c.sum(axis=0, dtype=dt)
            d_out = d_prev + d_c @ W
        return rec, kl, loss, {k: v.astype(dt) for k, v in g.items()}

    def train_step(self, x, eps, lr, kl_coeff, scales=None):
        rec, kl, loss, g = self.gradients(x, eps, kl_coeff, scales)
        self.adamw(g, lr)
        for n in self.enc_weight_names():                                           # vae_model.py:622-625
            np.maximum(self.p[n], self.dt(0), out=self.p[n])
        return rec, kl, loss
