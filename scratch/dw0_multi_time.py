"""Time of ontovae_dw0_global (ov_dw0_multi_kernel) at R simulated ranks x 128 rows: python scratch/dw0_multi_time.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_model, synth_X
from onto_vae_amd.vae_model import FusedAdamW
dev = torch.device("cuda", 0)
model, _ = build_model(dev)
e = model._engine
L, G = e.plan.L, e.plan.G
opt = FusedAdamW(model, lr=1e-4)
e.ensure_grad_buffer()
X = synth_X(8192, G, dev, seed=5)
e.ctx(128)
for R in (1, 2, 4, 8):
    blm = 128
    rstride = (2 * L * blm + 3 * L + 3) // 4 * 4
    gathered = torch.randn(R * rstride, device=dev) * 1e-3
    idx = torch.randperm(8192, device=dev)[:R * blm].to(torch.int32)
    colmap = torch.arange(R * blm, device=dev, dtype=torch.int32)
    hyper = opt.next_hyper()
    for _ in range(3):
        e.dw0_global(X, idx, colmap, R, blm, gathered, rstride, hyper)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        e.dw0_global(X, idx, colmap, R, blm, gathered, rstride, hyper)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 50
    print(f"R {R} K {R * blm}: {us:.1f} us, {2.0 * L * G * R * blm / us / 1e6:.1f} TFLOP/s")
