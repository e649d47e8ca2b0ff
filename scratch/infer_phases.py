"""Per-phase times of one inference pass (pass_data, 'act' output) at a large batch: python scratch/infer_phases.py [flags] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_model, synth_X
from onto_vae_amd import _lib
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
model, _ = build_model(dev)
e = model._engine
e.flags = flags
e.invalidate()
model.eval()
X = synth_X(B, e.plan.G, dev, seed=3)
out = torch.empty(B, e.plan.T, device=dev)
for _ in range(5):
    e.pass_data(X, None, None, None, act_out=out)
torch.cuda.synchronize()
L = _lib.lib()
nph = len(_lib.PHASES)
n = 20
_lib.check(L.ontovae_profile_begin((1 << nph) - 1, 64 * n), "begin")
for _ in range(n):
    e.pass_data(X, None, None, None, act_out=out)
torch.cuda.synchronize()
buf = (C.c_float * (64 * n))()
cnt = C.c_int32()
tot = 0
for ph, name in enumerate(_lib.PHASES):
    _lib.check(L.ontovae_profile_read(ph, buf, 64 * n, C.byref(cnt)), "read")
    if cnt.value:
        us = sum(buf[i] for i in range(cnt.value)) * 1e3 / n
        tot += us
        print(f"{name:16s} {us:8.1f} us/pass ({cnt.value // n} launches)")
L.ontovae_profile_end()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(n):
    e.pass_data(X, None, None, None, act_out=out)
ev[1].record()
torch.cuda.synchronize()
print(f"flags {flags:#x} B {B}: phases {tot:.1f} us, pass {ev[0].elapsed_time(ev[1]) * 1e3 / n:.1f} us")
