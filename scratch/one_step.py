import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from onto_vae_amd.vae_model import FusedAdamW
dev = torch.device('cuda',0)
model, masks = bench.build_model(dev)
e = model._engine
e.flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0; e.invalidate()
X = bench.synth_X(4096, 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
idx = torch.arange(128, dtype=torch.int32, device=dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
torch.cuda.synchronize()
