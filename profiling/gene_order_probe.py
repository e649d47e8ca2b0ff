"""Gene ordering on a REAL annotation: the Pathway Ontology of tests/golden/pwo_g2.npz with its genes in
the reference's order (sorted by name, ontobj.py:323) against the same genes clustered by shared parent terms (genes sorted by
their parent-term sets, so that genes below the same terms are neighbours: neighbouring rows of D / x / the gene mask).  The
permutation is applied to the mask rows here, outside the library (a model cannot tell: the loss is permutation-invariant), and
the two gene-layer kernels are timed with HIP events by the library's own profiler.  usage: python3 profiling/gene_order_probe.py [batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from onto_vae_amd import _lib
from onto_vae_amd.synthetic import StubOntobj, random_state_dict
from onto_vae_amd.vae_model import FusedAdamW, OntoVAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "pwo_g2.npz")))
masks = []
for i in range(int(fx["nmask"])):
    shape = tuple(int(v) for v in fx[f"mask_shape/{i}"])
    masks.append(np.unpackbits(fx[f"mask_bits/{i}"])[:shape[0] * shape[1]].reshape(shape).astype(np.int64))
G = masks[-1].shape[0]
gm = masks[-1]
# clustered order: lexicographic order of the parent-term sets (as sorted tuples), ties by name order
keys = [tuple(np.nonzero(gm[g])[0].tolist()) for g in range(G)]
clustered = np.array(sorted(range(G), key=lambda g: keys[g]))
# how alike neighbours are: mean Jaccard similarity of the parent sets of genes g and g + 1
def jaccard(order):
    s = 0.0
    for a, b in zip(order[:-1], order[1:]):
        A, Bs = set(keys[a]), set(keys[b])
        s += len(A & Bs) / max(len(A | Bs), 1)
    return s / (len(order) - 1)
print(f"PWO: {G} genes, {int(gm.sum())} gene-term links; neighbour similarity (Jaccard of parent sets): reference order "
      f"{jaccard(np.arange(G)):.3f}, clustered {jaccard(clustered):.3f}")
L = _lib.lib()
dev = torch.device("cuda", 0)
rng = np.random.RandomState(1)
X = torch.as_tensor(np.log1p(rng.poisson(0.3, size=(max(4 * B, 4096), G))).astype(np.float32)).to(dev)
for name, order in (("reference", np.arange(G)), ("clustered", clustered)):
    ms = masks[:-1] + [gm[order]]
    onto = StubOntobj(ms, data={"d": np.zeros((2, G))})
    torch.manual_seed(0)
    model = OntoVAE(onto, "d", neuronnum=3, drop=0.0, z_drop=0.0)
    e = model._engine
    Xo = X[:, torch.as_tensor(order).to(dev)].contiguous()
    opt = FusedAdamW(model, lr=1e-4)
    loss = torch.zeros(3, device=dev)
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    for _ in range(10):
        e.train_step(Xo, idx, None, 1e-4, opt.next_hyper(), loss)
    torch.cuda.synchronize()
    n = 50
    nph = len(_lib.PHASES)
    _lib.check(L.ontovae_profile_begin((1 << nph) - 1, 64 * n), "profile_begin")
    for _ in range(n):
        e.train_step(Xo, idx, None, 1e-4, opt.next_hyper(), loss)
    torch.cuda.synchronize()
    buf, cnt, out = (C.c_float * (64 * n))(), C.c_int32(), {}
    for ph, pname in enumerate(_lib.PHASES):
        _lib.check(L.ontovae_profile_read(ph, buf, 64 * n, C.byref(cnt)), "profile_read")
        out[pname] = sum(buf[i] for i in range(cnt.value)) / n * 1e3
    L.ontovae_profile_end()
    print(f"{name:10s} batch {B}: gene forward {out['dec_fwd_gene']:.2f} us, gene backward {out['dec_bwd_gene']:.2f} us, "
          f"step (sum of brackets) {sum(out.values()):.1f} us, loss {float(loss[2]):.1f}")
