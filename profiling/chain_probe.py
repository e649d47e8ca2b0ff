"""Where the time of the two chain kernels goes: the training step of the bench model with the chain's measurement switches
(bits 8-10 of ctx.flags in a -DOV_MEASUREMENT_SWITCHES build: 1 = no record requests / no epilogue, 2 = no slice work,
4 = no level barriers).  Results are wrong with a switch on; only the phase times are read.
    ONTOVAE_HIP_LIB=profiling/_dbg/libontovae_hip.so python3 profiling/chain_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from onto_vae_amd import _lib
from onto_vae_amd.vae_model import FusedAdamW

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model, masks = bench.build_model(dev)
e = model._engine
X = bench.synth_X(max(4096, B), 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
idx = torch.arange(B, dtype=torch.int32, device=dev)
L = _lib.lib()
n = 30
for dbg in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,4,3,7".split(","))]:
    e.flags = dbg << 8
    e.invalidate()
    for _ in range(5):
        e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    nph = len(_lib.PHASES)
    _lib.check(L.ontovae_profile_begin((1 << nph) - 1, 64 * n), "profile_begin")
    for _ in range(n):
        e.train_step(X, idx, None, 1e-4, opt.next_hyper(), loss)
    torch.cuda.synchronize()
    buf = (C.c_float * (64 * n))()
    cnt = C.c_int32()
    out = {}
    for ph, name in enumerate(_lib.PHASES):
        _lib.check(L.ontovae_profile_read(ph, buf, 64 * n, C.byref(cnt)), "profile_read")
        out[name] = sum(buf[i] for i in range(cnt.value)) / n * 1e3
    L.ontovae_profile_end()
    print({k: round(v, 1) for k, v in out.items() if v > 0}); print(f"dbg={dbg}: chain fwd {out['dec_fwd_terms']:.1f} us, chain bwd {out['dec_bwd']:.1f} us, total {sum(out.values()):.1f} us", flush=True)
