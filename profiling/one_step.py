"""A few training steps of the bench model and nothing else: the program rocprofv3's --pmc passes wrap (one counter group
per run; see profiles/*/README.md).  usage: python3 profiling/one_step.py [steps] [batch] [mode: train|infer] [ctx flags, e.g. 16 = no riders]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from onto_vae_amd.vae_model import FusedAdamW

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
dev = torch.device("cuda", 0)
model, masks = bench.build_model(dev)
e = model._engine
if len(sys.argv) > 4:
    e.flags = int(sys.argv[4])
    e.invalidate()
X = bench.synth_X(max(4096, B), 20000, dev)
if mode == "train":
    opt = FusedAdamW(model, lr=1e-4)
    loss = torch.zeros(3, device=dev)
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    for _ in range(steps):
        e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
else:
    model.eval()
    out = torch.empty(B, e.plan.T, device=dev)
    for _ in range(steps):
        e.pass_data(X[:B], None, None, None, act_out=out)
torch.cuda.synchronize()
