import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
import bench
from onto_vae_amd import _lib
from onto_vae_amd.vae_model import FusedAdamW
dev = torch.device('cuda',0)
model, masks = bench.build_model(dev)
e = model._engine
X = bench.synth_X(4096, 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
idx = torch.arange(128, dtype=torch.int32, device=dev)
L = _lib.lib()
def run(flags, n=30):
    e.flags = flags; e.invalidate()
    for _ in range(5): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    nph = len(_lib.PHASES)
    _lib.check(L.ontovae_profile_begin((1<<nph)-1, 64*n), 'pb')
    for _ in range(n): e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    torch.cuda.synchronize()
    buf = (C.c_float*(64*n))(); cnt = C.c_int32(); out = {}
    for ph, name in enumerate(_lib.PHASES):
        L.ontovae_profile_read(ph, buf, 64*n, C.byref(cnt)); out[name] = round(sum(buf[i] for i in range(cnt.value))/n*1e3,1)
    L.ontovae_profile_end()
    return out
for fl in [int(a,0) for a in sys.argv[1:]] or [0]:
    r = run(fl); print(hex(fl), r, 'sum', round(sum(r.values()),1))
