"""get_pathway_activities end to end from a HOST dataset (PCIe-inclusive rate): python scratch/pcie_infer.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bench import build_model
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda", 0)
model, _ = build_model(dev)
e = model._engine
rng = np.random.default_rng(0)
X = np.log1p(rng.poisson(0.5, size=(rows, e.plan.G))).astype(np.float32)
for pinned in (False, True):
    data = torch.as_tensor(X)
    if pinned:
        data = data.pin_memory()
    model._pass_data(data[:4096], 'act')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model._pass_data(data, 'act')
    dt = time.perf_counter() - t0
    print(f"rows {rows} pinned {pinned}: {dt * 1e3:.1f} ms, {rows / dt / 1e3:.1f} k samples/s, H2D {rows * e.plan.G * 4 / dt / 1e9:.1f} GB/s", out.shape)
