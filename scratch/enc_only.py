"""Only the inference pass at one batch size (default 4096), few iterations, for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_model, synth_X
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
model, _ = build_model(dev)
e = model._engine
model.eval()
X = synth_X(B, e.plan.G, dev, seed=3)
out = torch.empty(B, e.plan.T, device=dev)
for _ in range(8):
    e.pass_data(X, None, None, None, act_out=out)
torch.cuda.synchronize()
