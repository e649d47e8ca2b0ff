import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from onto_vae_amd import _lib
from onto_vae_amd.vae_model import FusedAdamW
dev = torch.device('cuda',0)
model, masks = bench.build_model(dev)
e = model._engine
e.flags = 0; e.invalidate()
X = bench.synth_X(4096, 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
idx = torch.arange(128, dtype=torch.int32, device=dev)
for _ in range(10): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (2*4*16*32))()
assert L.ontovae_debug_chain_probe(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 4, 16, 32).astype(np.int64)
print('levels fwd', list(e.plan.sell_fwd_levels) if hasattr(e.plan, 'sell_fwd_levels') else None)
for d, name in enumerate(['fwd', 'bwd']):
    blk = t[d, 0]                      # workgroup 0
    t0 = blk[:, 0].min()
    print(name, 'slots: 0 start, 1 prologue issued, 2 prologue barrier, 2+l level-l barrier passed, 30 chain done, 31 end')
    for slot in range(32):
        col = blk[:, slot]
        if (col == 0).all(): continue
        v = col[col != 0] - t0
        print(f'  slot {slot:2d}: n {len(v):2d} min {v.min():7d} mean {v.mean():9.0f} max {v.max():7d}')
