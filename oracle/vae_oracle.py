"""CPU oracle for the plain VAE sibling (reference onto_vae/vae_model.py:913-1140).  TEST INFRASTRUCTURE ONLY.

Dense numpy restatement with a hand-written backward, like oracle/ontovae_oracle.py (same AdamW / BatchNorm arithmetic,
imported from there); pinned by tests/golden/vae_small.npz, produced by the imported reference (tools/gen_golden.py).

Reference map:
  Encoder   onto_vae/modules.py:33-84    [Linear, BatchNorm1d, Dropout, ReLU] per entry of layer_dims_enc, then mu / logvar heads
  Decoder   onto_vae/modules.py:112-154  the same blocks per entry of layer_dims_dec, then Linear(layer_dims_dec[-1], in_features)
  forward   onto_vae/vae_model.py:979-990;  vae_loss :995-1004;  train_round body :1026-1037;  val_round body :1061-1065
eps supplied.  Dropout: off unless the caller hands in keep-scales (`scales`: {block prefix: [B, width], "mu": [B, Z],
"logvar": [B, Z]}, entries 0 or 1 / (1 - p), nn.Dropout's inverted form at modules.py:48, 58, 63, 123), e.g. those of the HIP
path's Philox stream restated by oracle/philox_oracle.py.
"""
import math

import numpy as np

from ontovae_oracle import ADAM_B1, ADAM_B2, ADAM_EPS, ADAM_WD, BN_EPS, BN_MOMENTUM


class OracleVAE:
    def __init__(self, in_features, layer_dims_enc=(1000,), layer_dims_dec=(1000,), latent_dim=128, dtype=np.float32):
        self.dt = np.dtype(dtype).type
        self.G, self.Z = int(in_features), int(latent_dim)
        self.enc, self.dec = [int(v) for v in layer_dims_enc], [int(v) for v in layer_dims_dec]
        self.p, self.buf, self.opt = {}, {}, None

    # ------------------------------------------------------------------ names (the reference's state_dict keys)
    def blocks(self):
        """(prefix, in, out) of every [Linear, BatchNorm] block, encoder first."""
        out, cin = [], self.G
        for i, w in enumerate(self.enc):
            out.append((f"encoder.encoder.{i}", cin, w))
            cin = w
        cin = self.Z
        for i, w in enumerate(self.dec):
            out.append((f"decoder.decoder.{i}", cin, w))
            cin = w
        return out

    def param_names(self):
        names = []
        for pre, _, _ in self.blocks():
            names += [pre + ".0.weight", pre + ".0.bias", pre + ".1.weight", pre + ".1.bias"]
        names += ["encoder.mu.0.weight", "encoder.mu.0.bias", "encoder.logvar.0.weight", "encoder.logvar.0.bias"]
        n = len(self.dec)
        names += [f"decoder.decoder.{n}.0.weight", f"decoder.decoder.{n}.0.bias"]
        return names

    def load_state_dict(self, sd):
        for n in self.param_names():
            self.p[n] = np.array(sd[n], dtype=self.dt, copy=True)
        for pre, _, _ in self.blocks():
            self.buf[pre + ".1.running_mean"] = np.array(sd[pre + ".1.running_mean"], dtype=self.dt, copy=True)
            self.buf[pre + ".1.running_var"] = np.array(sd[pre + ".1.running_var"], dtype=self.dt, copy=True)
            self.buf[pre + ".1.num_batches_tracked"] = int(np.asarray(sd[pre + ".1.num_batches_tracked"]))

    def state_dict(self):
        sd = {k: v.copy() for k, v in self.p.items()}
        for k, v in self.buf.items():
            sd[k] = np.int64(v) if k.endswith("num_batches_tracked") else v.copy()
        return sd

    # ------------------------------------------------------------------ forward
    def _block(self, pre, a, train, cache, scale=None):
        p, dt = self.p, self.dt
        h = a @ p[pre + ".0.weight"].T + p[pre + ".0.bias"]
        if train:
            B = a.shape[0]
            mean = h.mean(axis=0, dtype=dt)
            var = ((h - mean) ** 2).mean(axis=0, dtype=dt)
            m = dt(BN_MOMENTUM)
            self.buf[pre + ".1.running_mean"] = (dt(1) - m) * self.buf[pre + ".1.running_mean"] + m * mean
            self.buf[pre + ".1.running_var"] = (dt(1) - m) * self.buf[pre + ".1.running_var"] + m * (var * dt(B / (B - 1)))
            self.buf[pre + ".1.num_batches_tracked"] += 1
        else:
            mean, var = self.buf[pre + ".1.running_mean"], self.buf[pre + ".1.running_var"]
        invstd = (dt(1) / np.sqrt(var + dt(BN_EPS))).astype(dt)
        xhat = (h - mean) * invstd
        y = xhat * p[pre + ".1.weight"] + p[pre + ".1.bias"]
        out = np.maximum(y if scale is None else y * scale, dt(0))          # Dropout, then ReLU
        cache.append(dict(pre=pre, a_in=a, xhat=xhat, invstd=invstd, y=y, scale=scale))
        return out

    def forward(self, x, eps, train, cache=None, scales=None):
        p, dt = self.p, self.dt
        cache = [] if cache is None else cache
        sc = (lambda k: None) if scales is None else (lambda k: np.asarray(scales[k], dtype=dt))
        a = np.asarray(x, dtype=dt)
        for i in range(len(self.enc)):
            a = self._block(f"encoder.encoder.{i}", a, train, cache, sc(f"encoder.encoder.{i}"))
        mu = a @ p["encoder.mu.0.weight"].T + p["encoder.mu.0.bias"]
        lv = a @ p["encoder.logvar.0.weight"].T + p["encoder.logvar.0.bias"]
        if scales is not None:
            mu, lv = mu * sc("mu"), lv * sc("logvar")
        z = (mu + np.asarray(eps, dtype=dt) * np.exp(dt(0.5) * lv)).astype(dt)
        top = a
        a = z
        for i in range(len(self.dec)):
            a = self._block(f"decoder.decoder.{i}", a, train, cache, sc(f"decoder.decoder.{i}"))
        n = len(self.dec)
        recon = a @ p[f"decoder.decoder.{n}.0.weight"].T + p[f"decoder.decoder.{n}.0.bias"]
        return recon.astype(dt), mu.astype(dt), lv.astype(dt), dict(top=top, z=z, last=a, cache=cache)

    def vae_loss(self, recon, mu, lv, x, kl_coeff):
        dt = self.dt
        kl = dt(-0.5) * np.sum(dt(1) + lv - mu * mu - np.exp(lv), dtype=dt)
        rec = np.sum((recon - x) ** 2, dtype=dt)
        return dt(rec), dt(kl), dt(rec + dt(kl_coeff) * kl)

    # ------------------------------------------------------------------ backward
    def _block_bwd(self, c, d_out, g):
        p, dt = self.p, self.dt
        pre = c["pre"]
        B = d_out.shape[0]
        d_y = d_out * (c["y"] > 0)
        if c["scale"] is not None:
            d_y = d_y * c["scale"]
        xhat, invstd = c["xhat"], c["invstd"]
        g[pre + ".1.weight"] = (d_y * xhat).sum(axis=0, dtype=dt)
        g[pre + ".1.bias"] = d_y.sum(axis=0, dtype=dt)
        d_xhat = d_y * p[pre + ".1.weight"]
        s1 = d_xhat.sum(axis=0, dtype=dt)
        s2 = (d_xhat * xhat).sum(axis=0, dtype=dt)
        d_h = (invstd / dt(B)) * (dt(B) * d_xhat - s1 - xhat * s2)
        g[pre + ".0.weight"] = d_h.T @ c["a_in"]
        g[pre + ".0.bias"] = d_h.sum(axis=0, dtype=dt)
        return d_h @ p[pre + ".0.weight"]

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
        cache = f["cache"]
        for c in reversed(cache[len(self.enc):]):
            d = self._block_bwd(c, d, g)
        kc = dt(kl_coeff)
        sigma = np.exp(dt(0.5) * lv)
        d_mu = d + kc * mu
        d_lv = d * eps * dt(0.5) * sigma + kc * dt(0.5) * (np.exp(lv) - dt(1))
        if scales is not None:
            d_mu, d_lv = d_mu * np.asarray(scales["mu"], dtype=dt), d_lv * np.asarray(scales["logvar"], dtype=dt)
        g["encoder.mu.0.weight"] = d_mu.T @ f["top"]
        g["encoder.mu.0.bias"] = d_mu.sum(axis=0, dtype=dt)
        g["encoder.logvar.0.weight"] = d_lv.T @ f["top"]
        g["encoder.logvar.0.bias"] = d_lv.sum(axis=0, dtype=dt)
        d = d_mu @ p["encoder.mu.0.weight"] + d_lv @ p["encoder.logvar.0.weight"]
        for c in reversed(cache[:len(self.enc)]):
            d = self._block_bwd(c, d, g)
        return rec, kl, loss, {k: v.astype(dt) for k, v in g.items()}

    def adamw(self, grads, lr):
        dt = self.dt
        if self.opt is None:
            self.opt = {"step": 0, "m": {n: np.zeros_like(v) for n, v in self.p.items()}, "v": {n: np.zeros_like(v) for n, v in self.p.items()}}
        o = self.opt
        o["step"] += 1
        t = o["step"]
        step_size = dt(lr / (1.0 - ADAM_B1 ** t))
        bc2_sqrt = dt(math.sqrt(1.0 - ADAM_B2 ** t))
        for n in self.param_names():
            w, gr, m, v = self.p[n], grads[n], o["m"][n], o["v"][n]
            w *= dt(1.0 - lr * ADAM_WD)
            m += (gr - m) * dt(1.0 - ADAM_B1)
            v *= dt(ADAM_B2)
            v += dt(1.0 - ADAM_B2) * gr * gr
            w -= step_size * (m / (np.sqrt(v) / bc2_sqrt + dt(ADAM_EPS)))

    def train_step(self, x, eps, lr, kl_coeff, scales=None):
        rec, kl, loss, g = self.gradients(x, eps, kl_coeff, scales)
        self.adamw(g, lr)
        return rec, kl, loss

    def val_step(self, x, eps, kl_coeff):
        x = np.asarray(x, dtype=self.dt)
        recon, mu, lv, _ = self.forward(x, eps, False)
        return self.vae_loss(recon, mu, lv, x, kl_coeff)
