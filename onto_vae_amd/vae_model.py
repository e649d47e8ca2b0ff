"""OntoVAE on MI355X: the public class surface of reference onto_vae/vae_model.py:18-457.

Constructor, method names, arguments, return types, printed strings, error messages and the
state_dict layout follow the reference; the arithmetic runs in libontovae_hip.so.
Differences that a caller can observe are listed in INTEGRATION.md (no autograd graph on the
returned tensors; eps and dropout masks come from a Philox stream, not from torch's generator;
losses are fetched once per epoch instead of once per step).
"""
import numpy as np
import torch
import torch.nn as nn

from .engine import Engine, ADAM_DEFAULTS
from .fast_data_loader import FastTensorDataLoader
from .modules import Encoder, OntoDecoder
from .packing import MaskPlan

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **kw):
        return it


class FusedAdamW:
    """torch.optim.AdamW(model.parameters(), lr) of reference vae_model.py:251, as state for the
    fused HIP update: flat exp_avg / exp_avg_sq buffers in the engine and a step counter."""

    def __init__(self, model, lr=1e-3, betas=ADAM_DEFAULTS["betas"], eps=ADAM_DEFAULTS["eps"],
                 weight_decay=ADAM_DEFAULTS["weight_decay"]):
        self.engine = model._engine
        self.engine.ensure_opt_state()
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.step_count = 0

    def next_hyper(self):
        self.step_count += 1
        g = self.param_groups[0]
        return Engine.adamw_struct(g["lr"], self.step_count, g["betas"], g["eps"], g["weight_decay"])

    def zero_grad(self, set_to_none=True):
        pass

    def state_dict(self):
        e = self.engine
        # the flat moments follow the packed order of the plan (ontovae_plan_row_k), which changed between library ABIs:
        # the ABI version travels with them and load_state_dict refuses a state written in another order
        from ._lib import ABI_VERSION
        return {"state": {"step": self.step_count, "exp_avg": e.M.detach().clone(), "exp_avg_sq": e.V.detach().clone()},
                "param_groups": [dict(g) for g in self.param_groups], "format": "onto_vae_amd.flat.v2", "packed_order_abi": ABI_VERSION}

    def load_state_dict(self, sd):
        from ._lib import ABI_VERSION
        e = self.engine
        if sd.get("format") != "onto_vae_amd.flat.v2" or int(sd.get("packed_order_abi", -1)) != ABI_VERSION:
            raise ValueError(f"optimizer state in format {sd.get('format')!r} / packed order of ABI {sd.get('packed_order_abi')!r}: this "
                             f"library packs the decoder in the order of ABI {ABI_VERSION}; its flat moments cannot be taken over "
                             "(the model's state_dict is dense and loads in any version)")
        if tuple(sd["state"]["exp_avg"].shape) != tuple(e.P.shape):
            raise ValueError("optimizer state belongs to a model of another size")
        e.ensure_opt_state()
        e.M.copy_(sd["state"]["exp_avg"]); e.V.copy_(sd["state"]["exp_avg_sq"])
        self.step_count = int(sd["state"]["step"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]


def _as_fused(model, optimizer, lr):
    if isinstance(optimizer, FusedAdamW):
        return optimizer
    if optimizer is None:
        return FusedAdamW(model, lr=lr)
    if isinstance(optimizer, torch.optim.AdamW):   # the reference's call shape: follow its LIVE hyper-parameters
        if len(optimizer.param_groups) != 1:
            raise ValueError("the fused AdamW takes one parameter group (the reference builds one, vae_model.py:251)")
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize"):
            raise ValueError("amsgrad / maximize are not supported by the fused AdamW")
        f = getattr(optimizer, "_ontovae_fused", None)
        if f is None:
            f = FusedAdamW(model, lr=g["lr"])
            optimizer._ontovae_fused = f
        # re-read on every round: an LR scheduler or a manual change between rounds must take effect
        f.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
        return f
    raise TypeError("optimizer must be onto_vae_amd.vae_model.FusedAdamW (or a torch.optim.AdamW whose settings are adopted)")


class OntoVAE(nn.Module):
    """A dense encoder combined with an ontology-structured decoder (reference vae_model.py:18-78).

    Parameters are those of the reference: ontobj (anything exposing .description, .genes,
    .masks, .data and, for terms=, .annot), dataset, top_thresh, bottom_thresh, neuronnum,
    drop, z_drop.
    """

    def __init__(self, ontobj, dataset, top_thresh=1000, bottom_thresh=30, neuronnum=3, drop=0.2, z_drop=0.5):
        super().__init__()
        key = str(top_thresh) + '_' + str(bottom_thresh)
        if key not in ontobj.genes.keys():
            raise ValueError('Available trimming thresholds are: ' + ', '.join(list(ontobj.genes.keys())))
        self.ontology = ontobj.description
        self.top = top_thresh
        self.bottom = bottom_thresh
        self.genes = ontobj.genes[key]
        self.in_features = len(self.genes)
        masks = ontobj.masks[key]['decoder']
        from .mask_builder import PackedMasks
        packed = isinstance(masks, PackedMasks)            # onto_vae_amd.mask_builder.create_masks: links, no dense masks
        shapes = masks.shapes if packed else [m.shape for m in masks]
        self.layer_dims_dec = np.array([shapes[0][1]] + [sh[0] for sh in shapes])
        self.latent_dim = self.layer_dims_dec[0] * neuronnum
        self.layer_dims_enc = [self.latent_dim]
        self.neuronnum = neuronnum
        self.drop = drop
        self.z_drop = z_drop
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self._mask_src = masks

        plan = MaskPlan(neuronnum=neuronnum, links=masks.links, shapes=masks.shapes) if packed else MaskPlan(masks, neuronnum)
        if plan.G != self.in_features:
            raise ValueError(f"the gene mask has {plan.G} rows but ontobj.genes lists {self.in_features} genes")
        object.__setattr__(self, "_engine", Engine(plan, self.device, drop, z_drop))
        self.encoder = Encoder(self.in_features, self.latent_dim, self.layer_dims_enc, self.drop, self.z_drop,
                               engine=self._engine)
        self.decoder = OntoDecoder(self.in_features, self.layer_dims_dec, masks, self.latent_dim, self.neuronnum,
                                   engine=self._engine)

        if dataset not in ontobj.data[key].keys():
            raise ValueError('Available datasets are: ' + ', '.join(list(ontobj.data[key].keys())))
        self.X = ontobj.data[key][dataset]
        # test hook: callable(B) -> eps [B, L]; None = Philox stream inside the kernels
        self.eps_source = None
        self.inference_batch = 4096
        # perturbation_screen: first encoder layer of a perturbed batch from the cached product plus rank-1 terms
        # (False: the full pass over a copy of the data with the columns overwritten, as `perturbation` does)
        self.screen_rank1 = True

    # ------------------------------------------------------------------ plumbing
    def set_precision(self, mfma_dtype):
        """Not in the reference.  'f32' (default) or 'bf16': BASELINE.json configs[4], the two large dense products of a pass
        (x W0^T and its weight gradient) on bf16 MFMA with fp32 accumulation; parameters, optimizer and everything else stay
        float32.  Stated tolerance against the float32 path: loss 2e-2 relative per step, activities cosine >= 0.999."""
        self._engine.set_precision(mfma_dtype)
        return self

    @property
    def mask_list(self):
        """vae_model.py:51-52: the masks as float32 tensors (built on access, they are dense)."""
        return [torch.tensor(np.asarray(m), dtype=torch.float32) for m in self._mask_src]

    def _apply(self, fn, recurse=True):
        from .engine import norm_device
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._engine.device))
        if norm_device(probe.device) != self._engine.device:
            self._engine.to(probe.device)
            self.device = str(probe.device.type)
            self.encoder.rebind()
            self.decoder.rebind()
        if probe.dtype != torch.float32:
            raise TypeError("the HIP OntoVAE computes in float32 only")
        return self

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._engine.invalidate()

    def _eps(self, B):
        return None if self.eps_source is None else self.eps_source(B)

    # ------------------------------------------------------------------ reference methods
    def reparameterize(self, mu, log_var):
        """vae_model.py:80-93 on caller-supplied tensors (torch ops; the training path fuses this)."""
        sigma = torch.exp(0.5 * log_var)
        eps = self._eps(mu.shape[0])
        eps = torch.randn_like(sigma) if eps is None else torch.as_tensor(eps, dtype=sigma.dtype, device=sigma.device)
        return mu + eps * sigma

    def get_embedding(self, x):
        """vae_model.py:95-106."""
        mu, log_var = self.encoder(x)
        return self.reparameterize(mu, log_var)

    def forward(self, x):
        """vae_model.py:108-118.  Returns (reconstruction, mu, log_var), each [B, .] on the device."""
        x = torch.as_tensor(x)
        return self._engine.forward(x, eps=self._eps(x.shape[0]), train=self.training)

    def vae_loss(self, reconstruction, mu, log_var, data, kl_coeff, mode='train', run=None):
        """vae_model.py:120-129 on caller-supplied tensors (torch ops; the training path fuses this)."""
        kl_loss = -0.5 * torch.sum(1. + log_var - mu.pow(2) - log_var.exp())
        rec_loss = torch.sum((reconstruction - data) ** 2)
        if run is not None:
            run["metrics/" + mode + "/kl_loss"].log(kl_loss)
            run["metrics/" + mode + "/rec_loss"].log(rec_loss)
        return torch.mean(rec_loss + kl_coeff * kl_loss)

    def _batches(self, dataloader):
        """Yield (x, xidx) per batch: zero-copy (dataset, indices) for the HBM-resident loader,
        (batch, None) for any other iterable of tuples (vae_model.py:153-156)."""
        if isinstance(dataloader, FastTensorDataLoader) and dataloader.device == self._engine.device \
                and dataloader.tensors[0].dtype == torch.float32 and dataloader.tensors[0].is_contiguous():
            iter(dataloader)
            while True:
                idx = dataloader.next_indices()
                if idx is None:
                    return
                yield dataloader.tensors[0], idx
        else:
            for data in dataloader:
                yield data[0], None

    def _collect(self, losses, n, mode, run):
        """One device-to-host copy per round instead of loss.item() per step (vae_model.py:162)."""
        host = losses[:n].cpu().double().numpy()
        if run is not None:
            for rec, kl, _ in host:
                run["metrics/" + mode + "/kl_loss"].log(kl)
                run["metrics/" + mode + "/rec_loss"].log(rec)
        running_loss = 0.0
        for v in host[:, 2]:
            running_loss += float(v)
        return running_loss

    def train_round(self, dataloader, lr, kl_coeff, optimizer, run=None):
        """vae_model.py:131-180: one epoch of forward, ELBO, backward, masked AdamW, clamp."""
        self.train()
        opt = _as_fused(self, optimizer, lr)
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=self._engine.device)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            B = x.shape[0] if xidx is None else xidx.shape[0]
            if self._dp_world() > 1:
                from .parallel import dp_train_step
                dp_train_step(self, x, xidx, self._eps(B), kl_coeff, opt, losses[n], reduce_loss=False)
            else:
                self._engine.train_step(x, xidx, self._eps(B), kl_coeff, opt.next_hyper(), losses[n])
            n += 1
        if self._dp_world() > 1 and n > 0:
            from .parallel import all_reduce
            all_reduce(losses[:n])                          # the ranks' partial sums of the whole epoch, once
        return self._collect(losses, n, 'train', run) / len(dataloader)

    def val_round(self, dataloader, kl_coeff, run=None):
        """vae_model.py:182-213: eval-mode forward + loss (eps is still sampled)."""
        self.eval()
        losses = torch.zeros(max(len(dataloader), 1), 3, dtype=torch.float32, device=self._engine.device)
        n = 0
        for x, xidx in tqdm(self._batches(dataloader), total=len(dataloader), disable=None):
            B = x.shape[0] if xidx is None else xidx.shape[0]
            self._engine.eval_step(x, xidx, self._eps(B), kl_coeff, losses[n])
            n += 1
        return self._collect(losses, n, 'val', run) / len(dataloader)

    @staticmethod
    def _dp_world():
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def train_model(self, modelpath, lr=1e-4, kl_coeff=1e-4, batch_size=128, epochs=300, run=None, resume=None):
        """vae_model.py:215-273: 80/20 split (seed 42), AdamW defaults, best-val checkpointing.

        resume (not in the reference): path of a checkpoint written by this method; training
        continues after the epoch it holds, with its parameters, optimizer state, batch order and random streams, so
        that the continued run equals the uninterrupted one (with dropout off or on)."""
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
        first_epoch = 0
        if resume is not None:
            ck = torch.load(resume, map_location='cpu', weights_only=True)
            if 'resume' not in ck:
                raise ValueError("checkpoint holds no resume record (it was not written by onto_vae_amd)")
            self.load_state_dict(ck['model_state_dict'])
            optimizer.load_state_dict(ck['optimizer_state_dict'])
            r = ck['resume']
            trainloader.order = r['train_order'].to(dev)
            torch.set_rng_state(r['rng_state'])
            self._engine.rng_offset = int(r['rng_offset'])
            val_loss_min = float(ck['loss'])
            first_epoch = int(ck['epoch']) + 1

        for epoch in range(first_epoch, epochs):
            print(f"Epoch {epoch+1} of {epochs}")
            train_epoch_loss = self.train_round(trainloader, lr, kl_coeff, optimizer, run)
            val_epoch_loss = self.val_round(valloader, kl_coeff, run)

            if run is not None:
                run["metrics/train/loss"].log(train_epoch_loss)
                run["metrics/val/loss"].log(val_epoch_loss)

            if val_epoch_loss < val_loss_min:
                print('New best model!')
                if modelpath is not None and (self._dp_world() == 1 or torch.distributed.get_rank() == 0):
                    torch.save({
                        'epoch': epoch,
                        'model_state_dict': self.state_dict(),
                        'optimizer_state_dict': optimizer.state_dict(),
                        'loss': val_epoch_loss,
                        'resume': {'train_order': trainloader.order.cpu(), 'rng_state': torch.get_rng_state(),
                                   'rng_offset': int(self._engine.rng_offset)},
                    }, modelpath)
                val_loss_min = val_epoch_loss

            print(f"Train Loss: {train_epoch_loss:.4f}")
            print(f"Val Loss: {val_epoch_loss:.4f}")

    def _pass_data(self, data, output):
        """vae_model.py:276-327, streamed in batches: eval mode, two eps draws per sample, the
        nu-mean and the [z | activities] concatenation done on the device.

        The reference moves the whole dataset to the device in one tensor (vae_model.py:349).  Here a dataset that lives in
        host memory goes through a three-stage pipeline over three buffer sets: while the kernels of batch i run, batch i + 1
        is converted to float32 into pinned memory (by a few host threads) and copied up on a second stream, and the result of
        batch i - 1 is copied down on a third one into the (pinned) result array.  A dataset that is already a device tensor is read in place."""
        self.eval()
        e = self._engine
        e._require_gpu()
        T, G = e.plan.T, e.plan.G
        W = T if output == 'act' else G
        key = 'act_out' if output == 'act' else 'rec_out'
        step = max(int(self.inference_batch), 1)
        if isinstance(data, torch.Tensor) and data.device == e.device:
            data = data.to(torch.float32)
            N = data.shape[0]
            out = torch.empty(N, W, dtype=torch.float32, device=e.device)
            for i in range(0, N, step):
                xb = data[i:i + step]
                B = xb.shape[0]
                e.pass_data(xb, None, self._eps(B), self._eps(B), **{key: out[i:i + B]})
            return out.cpu().numpy()
        src = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data))
        if src.dim() != 2 or src.shape[1] != G:
            raise ValueError(f"expected data of shape [N, {G}], got {tuple(src.shape)}")
        N = src.shape[0]
        result = torch.empty(N, W, dtype=torch.float32, pin_memory=True)
        if N == 0:
            return result.numpy()
        nbuf = 3
        rows = min(step, N)
        # Staging into pinned memory is host work.  torch's copy_ converts float64 -> float32 on all its threads already
        # (measured on the GPU box: 8.0 ms per 4096 x 20000 batch, 123 GB/s of host memory traffic, against 5.7 ms for the copy
        # up: the conversion, not the bus, paces a float64 dataset at ~500 k samples/s); a same-dtype copy is ONE memcpy
        # thread (24 GB/s), so float32 sources are dealt to a few threads instead (copy_ releases the GIL).
        import concurrent.futures as _cf
        nthr = max(1, min(8, (torch.get_num_threads() or 1)))
        pool = _cf.ThreadPoolExecutor(nthr) if nthr > 1 and rows >= 64 and src.dtype == torch.float32 else None

        def stage(dst, lo, hi):
            if pool is None:
                dst[:hi - lo].copy_(src[lo:hi])
                return
            n = hi - lo
            per = (n + nthr - 1) // nthr
            futs = [pool.submit(lambda a, b: dst[a - lo:b - lo].copy_(src[a:b]), lo + j * per, min(lo + (j + 1) * per, hi))
                    for j in range(nthr) if lo + j * per < hi]
            for f in futs:
                f.result()

        with torch.cuda.device(e.device):
            main = torch.cuda.current_stream(e.device)
            s_in, s_out = torch.cuda.Stream(e.device), torch.cuda.Stream(e.device)
            hin = [torch.empty(rows, G, dtype=torch.float32, pin_memory=True) for _ in range(nbuf)]
            din = [torch.empty(rows, G, dtype=torch.float32, device=e.device) for _ in range(nbuf)]
            dout = [torch.empty(rows, W, dtype=torch.float32, device=e.device) for _ in range(nbuf)]
            up = [torch.cuda.Event() for _ in range(nbuf)]        # batch is in din[k]
            done = [torch.cuda.Event() for _ in range(nbuf)]      # kernels of the batch in slot k finished (din[k] free, dout[k] full)
            down = [torch.cuda.Event() for _ in range(nbuf)]      # dout[k] copied to the host (dout[k] free)
            used = [False] * nbuf
            for n, i in enumerate(range(0, N, step)):
                k = n % nbuf
                B = min(step, N - i)
                if used[k]:
                    up[k].synchronize()                           # hin[k] may be overwritten only after its copy went up
                stage(hin[k], i, i + B)                           # float64 -> float32 on the host cores, into pinned memory
                with torch.cuda.stream(s_in):
                    if used[k]:
                        s_in.wait_event(done[k])                  # the kernels that read din[k] three batches ago
                    din[k][:B].copy_(hin[k][:B], non_blocking=True)
                    up[k].record(s_in)
                main.wait_event(up[k])
                if used[k]:
                    main.wait_event(down[k])
                e.pass_data(din[k][:B], None, self._eps(B), self._eps(B), **{key: dout[k][:B]})
                done[k].record(main)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(done[k])
                    result[i:i + B].copy_(dout[k][:B], non_blocking=True)
                    down[k].record(s_out)
                used[k] = True
            s_out.synchronize()
        if pool is not None:
            pool.shutdown(wait=True)
        return result.numpy()

    # ---- shared pieces of the three retrieval methods --------------------------------------
    def _key(self):
        return str(self.top) + '_' + str(self.bottom)

    def _dataset_copy(self, ontobj, dataset, copy=True):
        """vae_model.py:343-346.  copy=False for the callers that only read the data (the reference copies there too, which
        nothing can observe: the pass never writes to it)."""
        if self.ontology != ontobj.description:
            raise ValueError('Wrong ontology provided, should be ' + self.ontology)
        data = ontobj.data[self._key()][dataset]
        return data.copy() if copy else data

    def _subset_terms(self, ontobj, res, terms):
        if terms is None:
            return res
        annot = ontobj.annot[self._key()]
        return res[:, annot[annot.ID.isin(terms)].index.to_numpy()]

    def _subset_genes(self, ontobj, res, rec_genes):
        if rec_genes is None:
            return res
        onto_genes = ontobj.genes[self._key()]
        return res[:, np.array([onto_genes.index(g) for g in rec_genes])]

    def get_pathway_activities(self, ontobj, dataset, terms=None):
        """vae_model.py:330-361: [N, T] float32 activities, columns ordered by depth."""
        data = self._dataset_copy(ontobj, dataset, copy=False)
        return self._subset_terms(ontobj, self._pass_data(data, 'act'), terms)

    def get_reconstructed_values(self, ontobj, dataset, rec_genes=None):
        """vae_model.py:364-395: [N, G] float32 reconstruction."""
        data = self._dataset_copy(ontobj, dataset, copy=False)
        return self._subset_genes(ontobj, self._pass_data(data, 'rec'), rec_genes)

    def perturbation(self, ontobj, dataset, genes, values, output='terms', terms=None, rec_genes=None):
        """vae_model.py:398-456: overwrite the listed gene columns with `values`, then the same
        pass as above ('terms' -> activities, 'genes' -> reconstruction)."""
        data = self._dataset_copy(ontobj, dataset)
        for g, v in zip(genes, values):
            data[:, self.genes.index(g)] = v
        if output not in ('terms', 'genes'):
            raise ValueError("output must be 'terms' or 'genes'")
        res = self._pass_data(data, 'act' if output == 'terms' else 'rec')
        return self._subset_genes(ontobj, self._subset_terms(ontobj, res, terms), rec_genes)

    def perturbation_screen(self, ontobj, dataset, perturbations, output='terms', direction='up', terms=None, rec_genes=None):
        """A screen of many perturbations of one dataset: for each (genes, values) pair what `perturbation`
        (vae_model.py:398-456) followed by Ontobj.wilcox_test against the unperturbed pass (ontobj.py:651-701) gives,
        with the data, both activity matrices and the tests resident on the device.

        The first encoder layer of the dataset is computed once; a perturbation adds one rank-1 term per gene to it
        (ontovae_pass_data_perturbed) instead of repeating the [N, G] x [G, L] product.  Lists of more than 64 genes
        take the ordinary pass over a copy of the data with the columns overwritten.

        Returns {'stat', 'pval', 'qval'}: float64 arrays [len(perturbations), C], C = the selected terms (output=
        'terms') or genes ('genes'), in the column order of get_pathway_activities / get_reconstructed_values; qval
        is the Benjamini-Hochberg correction within each perturbation, over the selected columns."""
        from . import stats as dstats
        if output not in ('terms', 'genes'):
            raise ValueError("output must be 'terms' or 'genes'")
        e = self._engine
        self.eval()
        X = torch.as_tensor(self._dataset_copy(ontobj, dataset), dtype=torch.float32).to(e.device).contiguous()
        N = X.shape[0]
        Cn = e.plan.T if output == 'terms' else e.plan.G
        step = int(self.inference_batch)
        ld = max((N + 3) // 4 * 4, 4)
        hpre = torch.empty(e.plan.L, ld if self.screen_rank1 else 4, dtype=torch.float32, device=e.device)
        stage = torch.empty(min(step, max(N, 1)), Cn, dtype=torch.float32, device=e.device)
        ctrl = torch.zeros(Cn, ld, dtype=torch.float32, device=e.device)
        pert = torch.zeros(Cn, ld, dtype=torch.float32, device=e.device)
        gene_pos = {g: i for i, g in enumerate(self.genes)}
        out_key = 'act_out' if output == 'terms' else 'rec_out'

        def run(dst, gidx, gval, data):
            for i in range(0, N, step):
                xb = data[i:i + step]
                B = xb.shape[0]
                kw = {out_key: stage[:B]}
                if gidx is None:
                    e.pass_data(xb, None, self._eps(B), self._eps(B), **kw)
                elif output == 'terms':          # activities straight into their columns of the feature-major matrix
                    e.pass_data_perturbed(xb, None, hpre, i, gidx, gval, self._eps(B), self._eps(B), act_fm=dst, act_fm_col0=i)
                    continue
                else:
                    e.pass_data_perturbed(xb, None, hpre, i, gidx, gval, self._eps(B), self._eps(B), **kw)
                dstats.to_feature_major(stage[:B], dst, i)

        rank1 = bool(self.screen_rank1)
        if rank1:
            for i in range(0, N, step):
                e.encode_pre(X[i:i + step], None, hpre, i)
            run(ctrl, np.zeros(0, np.int32), np.zeros(0, np.float32), X)
        else:
            run(ctrl, None, None, X)

        P = len(perturbations)
        stat = torch.empty(P, Cn, dtype=torch.float64, device=e.device)
        pval = torch.empty(P, Cn, dtype=torch.float64, device=e.device)
        alternative = 'greater' if direction == 'up' else 'less'
        for p, (genes, values) in enumerate(perturbations):
            last = {}
            for g, v in zip(genes, values):                  # sequential overwrite: the last value of a repeated gene stays
                if g not in gene_pos:
                    raise ValueError(f"{g!r} is not in list")
                last[gene_pos[g]] = float(v)
            gidx = np.fromiter(last.keys(), dtype=np.int32, count=len(last))
            gval = np.fromiter(last.values(), dtype=np.float32, count=len(last))
            if rank1 and len(gidx) <= 64:
                run(pert, gidx, gval, X)
            else:
                Xp = X.clone()
                Xp[:, torch.as_tensor(gidx, dtype=torch.int64, device=e.device)] = torch.as_tensor(gval, device=e.device)
                run(pert, None, None, Xp)
            dstats.wilcoxon_zsplit(pert, ctrl, N, alternative, out=(stat[p], pval[p]))

        cols = None
        if output == 'terms' and terms is not None:
            annot = ontobj.annot[self._key()]
            cols = annot[annot.ID.isin(terms)].index.to_numpy()
        if output == 'genes' and rec_genes is not None:
            cols = np.array([gene_pos[g] for g in rec_genes])
        if cols is not None:
            ci = torch.as_tensor(cols, dtype=torch.int64, device=e.device)
            stat, pval = stat[:, ci].contiguous(), pval[:, ci].contiguous()
        qval = dstats.fdr_bh(pval) if P and pval.shape[1] else torch.empty_like(pval)
        return {'stat': stat.cpu().numpy(), 'pval': pval.cpu().numpy(), 'qval': qval.cpu().numpy()}


from .dense_vae import VAE  # noqa: E402,F401  (reference vae_model.py:913: the plain VAE lives in the same module)


def __getattr__(name):
    """OntoEncVAE lives in vae_model.py upstream (reference vae_model.py:468); here it is imported on first use (its module
    imports this one)."""
    if name == "OntoEncVAE":
        from .enc_vae_model import OntoEncVAE
        return OntoEncVAE
    raise AttributeError(name)
