"""CPU oracle for the OntoVAE hot path.  TEST INFRASTRUCTURE ONLY.

This is a dense numpy (float32) restatement of the reference algorithm, written
from the reference's behaviour, with every gradient worked out by hand so that it
is independent of torch.autograd.  It exists to check the HIP path
(onto_vae_amd) and as the `cpu_baseline` leg of bench.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import it; the product
path never does (and fails loudly when the HIP library is missing).

Parity pin: the reference ships no tests or golden vectors, so
this oracle is pinned by fixtures generated from the *imported reference itself*
in the build container (tools/gen_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

It uses the reference's dense formulation on purpose: dense 0/1 masks expanded
with repeat_interleave, one dense Linear per depth with a prepending concat, a
dense gradient-mask multiply, dense AdamW over every parameter and a dense clamp.
That is what the reference costs on a CPU, so it is also the honest CPU baseline.

Reference map (paths relative to /root/reference):
  expand_masks            onto_vae/modules.py:268-275
  encoder_forward         onto_vae/modules.py:45-50, 56-64, 74-84
  decoder_forward         onto_vae/modules.py:303-313
  reparameterize          onto_vae/vae_model.py:91-93 (eps supplied by the caller)
  vae_loss                onto_vae/vae_model.py:124-129
  train_step              onto_vae/vae_model.py:156-176 (one loop body of train_round)
  val_step                onto_vae/vae_model.py:204-209
  pass_data               onto_vae/vae_model.py:276-327
  loader_batches          onto_vae/fast_data_loader.py:27-43
  train_val_split         onto_vae/vae_model.py:233-236
  AdamW arithmetic        torch.optim.AdamW defaults (vae_model.py:251): betas (0.9, 0.999),
                          eps 1e-8, weight_decay 1e-2, decoupled decay, bias correction
  BatchNorm1d arithmetic  torch.nn.BatchNorm1d defaults (modules.py:47): eps 1e-5, momentum 0.1,
                          biased variance for normalisation, unbiased for running_var
"""
import math
import time

import numpy as np

F32 = np.float32

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
ADAM_B1, ADAM_B2, ADAM_EPS, ADAM_WD = 0.9, 0.999, 1e-8, 1e-2


def expand_masks(mask_list, nu):
    """modules.py:271-275: inner masks repeat_interleave(nu) on both axes, last on columns only."""
    out = []
    for m in mask_list[:-1]:
        m = np.asarray(m)
        out.append(np.repeat(np.repeat(m, nu, axis=0), nu, axis=1).astype(F32))
    out.append(np.repeat(np.asarray(mask_list[-1]), nu, axis=1).astype(F32))
    return out


def nu_mean(a, nu):
    """vae_model.py:297/317: np.array(np.split(a, a.shape[1]/nu, axis=1)).mean(axis=2).T"""
    n, c = a.shape
    return a.reshape(n, c // nu, nu).mean(axis=2, dtype=a.dtype).astype(a.dtype)


def train_val_split(n):
    """vae_model.py:233-235."""
    idx = np.random.RandomState(seed=42).permutation(n)
    cut = round(len(idx) * 0.8)
    return idx[:cut], idx[cut:]


def loader_batches(n, batch_size, perms):
    """fast_data_loader.py:27-43 with the per-epoch permutations supplied.

    `perms` is a list with one permutation (or None for shuffle=False) per epoch.
    The shuffle is cumulative: epoch e sees tensors[r_1][r_2]...[r_e].
    Returns, per epoch, the list of index arrays into the ORIGINAL row order.
    """
    order = np.arange(n)
    epochs = []
    for r in perms:
        if r is not None:
            order = order[np.asarray(r)]
        epochs.append([order[i:i + batch_size].copy() for i in range(0, n, batch_size)])
    return epochs


class OracleOntoVAE:
    """Dense CPU OntoVAE, eps supplied; dropout off unless keep-scales are supplied (gradients / train_step).

    dtype=np.float64 evaluates the same algorithm in double precision: the yardstick that tells fp32 evaluation-order
    noise from errors in tests (never the thing compared against for parity; that is the float32 instance)."""

    def __init__(self, mask_list, neuronnum=3, dtype=np.float32):
        self.dt = np.dtype(dtype).type
        self.nu = int(neuronnum)
        self.mask_list = [np.asarray(m) for m in mask_list]
        self.K = len(mask_list)
        dims = [self.mask_list[0].shape[1]] + [m.shape[0] for m in self.mask_list]
        self.layer_dims_dec = np.array(dims)                       # vae_model.py:53
        self.latent_dim = int(dims[0] * self.nu)                   # vae_model.py:54
        self.in_features = int(dims[-1])
        d = np.hstack([np.array(dims[:-1]) * self.nu, dims[-1]])   # modules.py:268
        self.layer_shapes = [(int(np.sum(d[:i + 1])), int(d[i + 1])) for i in range(len(d) - 1)]
        self.masks = [m.astype(self.dt) for m in expand_masks(self.mask_list, self.nu)]
        for m, (cin, cout) in zip(self.masks, self.layer_shapes):
            assert m.shape == (cout, cin), (m.shape, cout, cin)
        self.p = {}          # parameters, state_dict names
        self.buf = {}        # BN buffers
        self.opt = None      # AdamW state
        self.training = True
        self.phase_seconds = None      # bench.py's cpu_baseline: dict that train_step fills (forward / backward / grad_mask / adamw / clamp)

    # ------------------------------------------------------------------ state
    def param_names(self):
        names = ["encoder.encoder.0.0.weight", "encoder.encoder.0.0.bias",
                 "encoder.encoder.0.1.weight", "encoder.encoder.0.1.bias",
                 "encoder.mu.0.weight", "encoder.mu.0.bias",
                 "encoder.logvar.0.weight", "encoder.logvar.0.bias"]
        for i in range(self.K):
            names += [f"decoder.decoder.{i}.0.weight", f"decoder.decoder.{i}.0.bias"]
        return names

    def load_state_dict(self, sd):
        for n in self.param_names():
            self.p[n] = np.array(sd[n], dtype=self.dt, copy=True)
        self.buf["running_mean"] = np.array(sd["encoder.encoder.0.1.running_mean"], dtype=self.dt, copy=True)
        self.buf["running_var"] = np.array(sd["encoder.encoder.0.1.running_var"], dtype=self.dt, copy=True)
        self.buf["num_batches_tracked"] = int(np.asarray(sd["encoder.encoder.0.1.num_batches_tracked"]))
        L, G = self.latent_dim, self.in_features
        assert self.p["encoder.encoder.0.0.weight"].shape == (L, G)
        for i, (cin, cout) in enumerate(self.layer_shapes):
            assert self.p[f"decoder.decoder.{i}.0.weight"].shape == (cout, cin)

    def state_dict(self):
        sd = {k: v.copy() for k, v in self.p.items()}
        sd["encoder.encoder.0.1.running_mean"] = self.buf["running_mean"].copy()
        sd["encoder.encoder.0.1.running_var"] = self.buf["running_var"].copy()
        sd["encoder.encoder.0.1.num_batches_tracked"] = np.int64(self.buf["num_batches_tracked"])
        return sd

    # ---------------------------------------------------------------- forward
    def encoder_forward(self, x, train, scales=None):
        """scales: None, or (hidden, mu, log_var) keep-scales [B, L] of the three Dropout modules (modules.py:48, 58, 63),
        values 0 or 1/(1-p); only legal in train mode (nn.Dropout is the identity in eval mode)."""
        p = self.p
        assert scales is None or train
        h = x @ p["encoder.encoder.0.0.weight"].T + p["encoder.encoder.0.0.bias"]
        if train:
            B = x.shape[0]
            if B < 2:
                raise ValueError("Expected more than 1 value per channel when training")
            mean = h.mean(axis=0, dtype=self.dt)
            var = ((h - mean) ** 2).mean(axis=0, dtype=self.dt)          # biased
            m = self.dt(BN_MOMENTUM)
            self.buf["running_mean"] = (self.dt(1) - m) * self.buf["running_mean"] + m * mean
            self.buf["running_var"] = (self.dt(1) - m) * self.buf["running_var"] + m * (var * self.dt(B / (B - 1)))
            self.buf["num_batches_tracked"] += 1
        else:
            mean, var = self.buf["running_mean"], self.buf["running_var"]
        invstd = (self.dt(1) / np.sqrt(var + self.dt(BN_EPS))).astype(self.dt)
        xhat = (h - mean) * invstd
        y = xhat * p["encoder.encoder.0.1.weight"] + p["encoder.encoder.0.1.bias"]
        if scales is not None:
            y = y * np.asarray(scales[0], dtype=self.dt)               # BN, then Dropout, then ReLU (modules.py:46-49)
        a = np.maximum(y, self.dt(0))
        mu = a @ p["encoder.mu.0.weight"].T + p["encoder.mu.0.bias"]
        lv = a @ p["encoder.logvar.0.weight"].T + p["encoder.logvar.0.bias"]
        if scales is not None:
            mu = mu * np.asarray(scales[1], dtype=self.dt)             # modules.py:56-64: Dropout(z_drop) after each head
            lv = lv * np.asarray(scales[2], dtype=self.dt)
        cache = dict(x=x, xhat=xhat, invstd=invstd, y=y, a=a)
        return mu.astype(self.dt), lv.astype(self.dt), cache

    def decoder_forward(self, z, collect=False, gene_layer=True):
        out = z
        outs, cs = [out], []
        for i in range(self.K - 1):
            c = out @ self.p[f"decoder.decoder.{i}.0.weight"].T + self.p[f"decoder.decoder.{i}.0.bias"]
            cs.append(c)
            out = np.concatenate((c, out), axis=1)                   # new block is PREPENDED
            outs.append(out)
        i = self.K - 1
        if not gene_layer:
            return None, outs, cs
        recon = out @ self.p[f"decoder.decoder.{i}.0.weight"].T + self.p[f"decoder.decoder.{i}.0.bias"]
        if collect:
            return recon.astype(self.dt), outs, cs
        return recon.astype(self.dt)

    def reparameterize(self, mu, lv, eps):
        return (mu + eps * np.exp(self.dt(0.5) * lv)).astype(self.dt)

    def forward(self, x, eps, train=None):
        train = self.training if train is None else train
        x = np.asarray(x, dtype=self.dt)
        mu, lv, _ = self.encoder_forward(x, train)
        z = self.reparameterize(mu, lv, np.asarray(eps, dtype=self.dt))
        return self.decoder_forward(z), mu, lv

    def vae_loss(self, recon, mu, lv, x, kl_coeff):
        kl = self.dt(-0.5) * np.sum(self.dt(1) + lv - mu * mu - np.exp(lv), dtype=self.dt)
        rec = np.sum((recon - x) ** 2, dtype=self.dt)
        return self.dt(rec), self.dt(kl), self.dt(rec + self.dt(kl_coeff) * kl)

    # --------------------------------------------------------------- training
    def _init_opt(self):
        self.opt = {"step": 0,
                    "m": {n: np.zeros_like(v) for n, v in self.p.items()},
                    "v": {n: np.zeros_like(v) for n, v in self.p.items()}}

    def gradients(self, x, eps, kl_coeff, scales=None):
        """Forward in train mode + hand-written backward.  Returns (rec, kl, loss, grads).
        scales: dropout keep-scales, see encoder_forward."""
        p, K = self.p, self.K
        x = np.asarray(x, dtype=self.dt)
        eps = np.asarray(eps, dtype=self.dt)
        B = x.shape[0]
        mu, lv, c = self.encoder_forward(x, True, scales)
        sigma = np.exp(self.dt(0.5) * lv)
        z = (mu + eps * sigma).astype(self.dt)
        recon, outs, _ = self.decoder_forward(z, collect=True)
        rec, kl, loss = self.vae_loss(recon, mu, lv, x, kl_coeff)
        if self.phase_seconds is not None:      # the encoder part of the forward was timed by train_step's bracket; see there
            self.phase_seconds["_fwd_end"] = time.perf_counter()

        g = {}
        d_c = self.dt(2) * (recon - x)                                  # d loss / d recon
        d_out = None
        for i in range(K - 1, -1, -1):
            W = p[f"decoder.decoder.{i}.0.weight"]
            if i < K - 1:
                n_c = W.shape[0]
                d_c, d_rest = d_out[:, :n_c], d_out[:, n_c:]
            g[f"decoder.decoder.{i}.0.weight"] = d_c.T @ outs[i]
            g[f"decoder.decoder.{i}.0.bias"] = d_c.sum(axis=0, dtype=self.dt)
            d_prev = d_c @ W
            d_out = d_prev if i == K - 1 else d_rest + d_prev
        d_z = d_out
        kc = self.dt(kl_coeff)
        d_mu = d_z + kc * mu
        d_lv = d_z * eps * self.dt(0.5) * sigma + kc * self.dt(0.5) * (np.exp(lv) - self.dt(1))
        if scales is not None:
            d_mu = d_mu * np.asarray(scales[1], dtype=self.dt)
            d_lv = d_lv * np.asarray(scales[2], dtype=self.dt)
        a = c["a"]
        g["encoder.mu.0.weight"] = d_mu.T @ a
        g["encoder.mu.0.bias"] = d_mu.sum(axis=0, dtype=self.dt)
        g["encoder.logvar.0.weight"] = d_lv.T @ a
        g["encoder.logvar.0.bias"] = d_lv.sum(axis=0, dtype=self.dt)
        d_a = d_mu @ p["encoder.mu.0.weight"] + d_lv @ p["encoder.logvar.0.weight"]
        d_y = d_a * (c["y"] > 0)
        if scales is not None:
            d_y = d_y * np.asarray(scales[0], dtype=self.dt)
        xhat, invstd = c["xhat"], c["invstd"]
        g["encoder.encoder.0.1.weight"] = (d_y * xhat).sum(axis=0, dtype=self.dt)
        g["encoder.encoder.0.1.bias"] = d_y.sum(axis=0, dtype=self.dt)
        d_xhat = d_y * p["encoder.encoder.0.1.weight"]
        s1 = d_xhat.sum(axis=0, dtype=self.dt)
        s2 = (d_xhat * xhat).sum(axis=0, dtype=self.dt)
        d_h = (invstd / self.dt(B)) * (self.dt(B) * d_xhat - s1 - xhat * s2)
        g["encoder.encoder.0.0.weight"] = d_h.T @ x
        g["encoder.encoder.0.0.bias"] = d_h.sum(axis=0, dtype=self.dt)
        g = {k: v.astype(self.dt) for k, v in g.items()}
        return rec, kl, loss, g

    def adamw(self, grads, lr):
        """torch.optim.AdamW, single-tensor arithmetic order, fp32 state."""
        if self.opt is None:
            self._init_opt()
        o = self.opt
        o["step"] += 1
        t = o["step"]
        bc1 = 1.0 - ADAM_B1 ** t
        bc2 = 1.0 - ADAM_B2 ** t
        step_size = self.dt(lr / bc1)
        bc2_sqrt = self.dt(math.sqrt(bc2))
        for n in self.param_names():
            w, gr, m, v = self.p[n], grads[n], o["m"][n], o["v"][n]
            w *= self.dt(1.0 - lr * ADAM_WD)
            m += (gr - m) * self.dt(1.0 - ADAM_B1)                       # lerp
            v *= self.dt(ADAM_B2)
            v += self.dt(1.0 - ADAM_B2) * gr * gr
            denom = np.sqrt(v) / bc2_sqrt + self.dt(ADAM_EPS)
            w -= step_size * (m / denom)

    def apply_step(self, g, lr):
        """Gradient mask, AdamW and clamp of one train_round body (vae_model.py:168-176) on gradients from gradients()."""
        ph, t0 = self.phase_seconds, time.perf_counter()
        for i in range(self.K):                                     # vae_model.py:168-169
            g[f"decoder.decoder.{i}.0.weight"] *= self.masks[i]
        t1 = time.perf_counter()
        self.adamw(g, lr)                                           # vae_model.py:172
        t2 = time.perf_counter()
        for i in range(self.K):                                     # vae_model.py:175-176
            n = f"decoder.decoder.{i}.0.weight"
            self.p[n] = np.maximum(self.p[n], self.dt(0))
        if ph is not None:
            for k, v in (("grad_mask", t1 - t0), ("adamw", t2 - t1), ("clamp", time.perf_counter() - t2)):
                ph[k] = ph.get(k, 0.0) + v

    def train_step(self, x, eps, lr, kl_coeff, scales=None):
        """One body of train_round (vae_model.py:156-176).  Returns (rec, kl, loss).  With self.phase_seconds a dict, the
        wall time of the five phases BASELINE.md's CPU protocol names is added to it: forward (encoder, decoder, loss),
        backward, grad_mask, adamw, clamp."""
        ph, t0 = self.phase_seconds, time.perf_counter()
        rec, kl, loss, g = self.gradients(x, eps, kl_coeff, scales)
        if ph is not None:
            t1, tf = time.perf_counter(), ph.pop("_fwd_end")
            ph["forward"] = ph.get("forward", 0.0) + (tf - t0)
            ph["backward"] = ph.get("backward", 0.0) + (t1 - tf)
        self.apply_step(g, lr)
        return rec, kl, loss

    def val_step(self, x, eps, kl_coeff):
        x = np.asarray(x, dtype=self.dt)
        recon, mu, lv = self.forward(x, eps, train=False)
        return self.vae_loss(recon, mu, lv, x, kl_coeff)

    # -------------------------------------------------------------- inference
    def pass_data(self, x, eps1, eps2, output):
        """vae_model.py:276-327: eval mode, two eps draws, hooks on the first K-1 layers.
        output 'act_only' is 'act' without evaluating the gene layer (whose result 'act' discards): for large test sizes."""
        x = np.asarray(x, dtype=self.dt)
        mu, lv, _ = self.encoder_forward(x, False)
        z1 = self.reparameterize(mu, lv, np.asarray(eps1, dtype=self.dt))       # get_embedding
        z_avg = nu_mean(z1, self.nu)
        z2 = self.reparameterize(mu, lv, np.asarray(eps2, dtype=self.dt))       # second forward
        recon, _, cs = self.decoder_forward(z2, collect=True, gene_layer=output != "act_only")
        if output == "rec":
            return recon
        act = np.concatenate(cs, axis=1) if cs else np.zeros((x.shape[0], 0), self.dt)
        return np.hstack((z_avg, nu_mean(act, self.nu))).astype(self.dt)
