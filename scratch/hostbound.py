import sys, time; sys.path.insert(0,'.')
import torch, bench
from onto_vae_amd.vae_model import FusedAdamW
dev = torch.device('cuda',0)
model, masks = bench.build_model(dev)
e = model._engine
X = bench.synth_X(4096, 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
idx = torch.arange(128, dtype=torch.int32, device=dev)
for fl in [int(a,0) for a in sys.argv[1:]] or [0]:
    e.flags = fl; e.invalidate()
    for _ in range(20): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(hex(fl), 'enqueue us/step', round((t1-t0)/n*1e6,1), 'total us/step', round((t2-t0)/n*1e6,1))
