"""Synthetic ontologies and expression data for tests and benchmarks.

The generators follow SURVEY.md section 8(d) draw for draw, so the counts quoted
there (4 595 term-term links, 159 732 gene-term links, 520 551 expanded
non-zeros for GOBP-syn) are reproduced on any machine from the numpy seed alone.

The mask format is the reference's input contract
(reference onto_vae/ontobj.py:327-392): a list of K int64 0/1 matrices, mask i
with rows = level i+1 and columns = [level i | level i-1 | ... | level 0],
genes appended as the deepest level.
"""
import numpy as np

GOBP_SYN_COUNTS = [120, 60, 150, 330, 480, 540, 480, 360, 240, 140, 70, 30]
GOBP_SYN_GENES = 20000
GOBP_SYN_SEED = 20250117

FULLGO_SYN_COUNTS = [200, 90, 220, 480, 700, 800, 720, 540, 360, 210, 100, 50, 20, 10]
FULLGO_SYN_SEED = 20250118


def synthetic_links(counts, n_genes, rng):
    """Draw term->parent and gene->parent links."""
    counts = list(counts)
    K = len(counts)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    T = int(off[-1])
    term_parents = [set() for _ in range(T)]
    for d in range(1, K):
        for t in range(off[d], off[d + 1]):
            ps = {int(rng.randint(off[d - 1], off[d]))}
            for _ in range(rng.poisson(0.6)):
                ps.add(int(rng.randint(0, off[d])))
            term_parents[t] = ps
    gene_parents = []
    for _ in range(n_genes):
        gene_parents.append(set(int(v) for v in rng.randint(0, T, size=1 + rng.poisson(7))))
    return off, term_parents, gene_parents


def links_to_masks(off, term_parents, gene_parents):
    """Dense int64 decoder masks in the reference layout (ontobj.py:386-392)."""
    K = len(off) - 1
    T = int(off[-1])
    G = len(gene_parents)

    def col_of(term, level_hi):
        # columns are [level_hi | level_hi-1 | ... | level 0], each in index order
        d = int(np.searchsorted(off, term, side="right") - 1)
        base = int(sum(off[j + 1] - off[j] for j in range(d + 1, level_hi + 1)))
        return base + term - int(off[d])

    masks = []
    for i in range(K - 1):
        rows = int(off[i + 2] - off[i + 1])
        cols = int(off[i + 1])
        m = np.zeros((rows, cols), dtype=np.int64)
        for r, t in enumerate(range(off[i + 1], off[i + 2])):
            for p in term_parents[t]:
                m[r, col_of(p, i)] = 1
        masks.append(m)
    m = np.zeros((G, T), dtype=np.int64)
    for g, ps in enumerate(gene_parents):
        for p in ps:
            m[g, col_of(p, K - 1)] = 1
    masks.append(m)
    return masks


def synthetic_masks(counts=GOBP_SYN_COUNTS, n_genes=GOBP_SYN_GENES, seed=GOBP_SYN_SEED, return_rng=False):
    rng = np.random.RandomState(seed)
    off, tp, gp = synthetic_links(counts, n_genes, rng)
    masks = links_to_masks(off, tp, gp)
    return (masks, rng) if return_rng else masks


def synthetic_expression(rng, n, n_genes):
    """X = log1p(Poisson(0.3)), float64, continuing the mask generator's rng."""
    return np.log1p(rng.poisson(0.3, size=(n, n_genes))).astype(np.float64)


class StubOntobj:
    """Duck-typed stand-in for reference onto_vae/ontobj.py:Ontobj.

    OntoVAE only reads .description, .genes[key], .masks[key]['decoder'],
    .data[key][name] and (with terms=) .annot[key] (reference
    onto_vae/vae_model.py:43-78, 355-359).
    """

    def __init__(self, masks, data=None, description="synthetic", top=1000, bottom=30, gene_names=None, term_ids=None):
        key = f"{top}_{bottom}"
        G = masks[-1].shape[0]
        T = masks[-1].shape[1]
        self.description = description
        self.genes = {key: list(gene_names) if gene_names is not None else [f"G{i:06d}" for i in range(G)]}
        self.masks = {key: {"decoder": masks}}
        self.data = {key: dict(data or {})}
        ids = list(term_ids) if term_ids is not None else [f"T:{i:07d}" for i in range(T)]
        try:
            import pandas as pd
            self.annot = {key: pd.DataFrame({"ID": ids})}
        except Exception:  # pandas is optional for the hot path
            self.annot = {key: None}

    def add_dataset(self, name, X, top=1000, bottom=30):
        self.data[f"{top}_{bottom}"][name] = X


def random_state_dict(mask_list, nu, seed, dense=True):
    """A state_dict (numpy float32) with the reference's key layout, drawn from a numpy seed.

    Weights follow nn.Linear's default range U(-1/sqrt(fan_in), 1/sqrt(fan_in)); decoder
    weights are masked and clamped at 0 like OntoDecoder.__init__ does (reference
    onto_vae/modules.py:291-296).  BN affine and running statistics are made non-trivial so
    that eval-mode parity exercises them.  Independent of torch's RNG on purpose: the same
    values can be injected into the reference (load_state_dict), the oracle and the HIP model.
    """
    rng = np.random.RandomState(seed)
    dims = [mask_list[0].shape[1]] + [m.shape[0] for m in mask_list]
    G = dims[-1]
    L = dims[0] * nu
    f32 = np.float32

    def lin(out_f, in_f):
        b = 1.0 / np.sqrt(in_f)
        return (rng.uniform(-b, b, size=(out_f, in_f)).astype(f32), rng.uniform(-b, b, size=(out_f,)).astype(f32))

    sd = {}
    sd["encoder.encoder.0.0.weight"], sd["encoder.encoder.0.0.bias"] = lin(L, G)
    sd["encoder.encoder.0.1.weight"] = rng.uniform(0.5, 1.5, size=L).astype(f32)
    sd["encoder.encoder.0.1.bias"] = rng.uniform(-0.2, 0.2, size=L).astype(f32)
    sd["encoder.encoder.0.1.running_mean"] = rng.uniform(-0.1, 0.1, size=L).astype(f32)
    sd["encoder.encoder.0.1.running_var"] = rng.uniform(0.5, 1.5, size=L).astype(f32)
    sd["encoder.encoder.0.1.num_batches_tracked"] = np.int64(0)
    sd["encoder.mu.0.weight"], sd["encoder.mu.0.bias"] = lin(L, L)
    sd["encoder.logvar.0.weight"], sd["encoder.logvar.0.bias"] = lin(L, L)
    K = len(mask_list)
    cin = L
    for i, m in enumerate(mask_list):
        last = i == K - 1
        cout = m.shape[0] if last else m.shape[0] * nu
        b = 1.0 / np.sqrt(cin)
        mm = np.repeat(np.asarray(m), nu, axis=1)
        if not last:
            mm = np.repeat(mm, nu, axis=0)
        if dense:
            w = rng.uniform(-b, b, size=(cout, cin)).astype(f32)
            w = np.maximum(w * mm.astype(f32), f32(0))
        else:  # only draw the in-mask values (large models); same distribution, different stream
            w = np.zeros((cout, cin), dtype=f32)
            nz = np.nonzero(mm)
            w[nz] = np.maximum(rng.uniform(-b, b, size=len(nz[0])).astype(f32), f32(0))
        sd[f"decoder.decoder.{i}.0.weight"] = w
        sd[f"decoder.decoder.{i}.0.bias"] = rng.uniform(-b, b, size=(cout,)).astype(f32)
        cin += cout
    return sd
