import time, torch, numpy as np, threading
rows, G = 4096, 20000
src = torch.from_numpy(np.random.rand(4 * rows, G))
pins = [torch.empty(rows, G, dtype=torch.float32, pin_memory=True) for _ in range(3)]
devs = [torch.empty(rows, G, dtype=torch.float32, device="cuda") for _ in range(3)]
s_in = torch.cuda.Stream()
torch.cuda.synchronize()
# (a) conversion loop alone, (b) H2D loop alone, (c) both at once (H2D of buffer k while converting into buffer k+1)
def conv(n):
    for i in range(n): pins[i % 3].copy_(src[(i % 4) * rows:(i % 4 + 1) * rows])
def h2d(n):
    with torch.cuda.stream(s_in):
        for i in range(n): devs[i % 3].copy_(pins[(i + 1) % 3], non_blocking=True)
    s_in.synchronize()
for name, f in (("convert alone", lambda: conv(16)), ("h2d alone", lambda: h2d(16))):
    f(); t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    print(f"{name}: {dt/16*1e3:.2f} ms per batch")
t0 = time.perf_counter()
th = threading.Thread(target=h2d, args=(16,)); th.start(); conv(16); th.join()
print(f"both at once: {(time.perf_counter()-t0)/16*1e3:.2f} ms per batch")
# the pipeline as _pass_data runs it, without kernels
ev = [torch.cuda.Event() for _ in range(3)]
t0 = time.perf_counter()
for i in range(16):
    k = i % 3
    if i >= 3: ev[k].synchronize()
    pins[k].copy_(src[(i % 4) * rows:(i % 4 + 1) * rows])
    with torch.cuda.stream(s_in):
        devs[k].copy_(pins[k], non_blocking=True); ev[k].record(s_in)
s_in.synchronize()
print(f"pipeline convert -> async h2d: {(time.perf_counter()-t0)/16*1e3:.2f} ms per batch")
import os
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
