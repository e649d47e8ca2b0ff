"""Find reads of uninitialised workspace: poison the caching allocator's pool with NaN, then run the small OntoEncVAE."""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
from test_oracle_encvae import load_encvae
from conftest import sub
from test_gpu_encvae import build, grads_by_name
def poison():
    junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(4)]
    small = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22) for _ in range(8)]
    torch.cuda.synchronize(); del junk, small
fx, masks, _ = load_encvae("encvae_small")
G = masks[0].shape[1]
rng = np.random.RandomState(3)
X = np.log1p(rng.poisson(0.8, size=(200, G))).astype(np.float32)
for flags in (0, 1):
  for B in (64, 32, 128, 40):
    poison()
    m, onto = build(masks, sub(fx, "sd0/"), X=X)
    m.encoder._engine.flags = flags; m.encoder._engine._ctx = None
    m.train()
    out = torch.zeros(3, device="cuda")
    m.train_step(torch.as_tensor(X[:B]), None, 1e-4, None, out, eps=None, apply_update=False)
    torch.cuda.synchronize()
    g = grads_by_name(m)
    print("flags", flags, "B", B, "loss", out.tolist(), {k: int((~np.isfinite(v)).sum()) for k, v in g.items() if not np.isfinite(v).all()}, flush=True)
    e = m.encoder._engine
    print("   plan C, L, G, K:", e.plan.C, e.plan.L, e.plan.G, e.plan.K, "layer_nodes", e.plan.layer_nodes)
