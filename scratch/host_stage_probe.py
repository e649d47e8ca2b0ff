import time, torch, numpy as np, concurrent.futures as cf
rows, G = 4096, 20000
src = torch.from_numpy(np.random.rand(4 * rows, G))
pin = torch.empty(rows, G, dtype=torch.float32, pin_memory=True)
dev = torch.empty(rows, G, dtype=torch.float32, device="cuda")
def t(f, n=8):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print("threads", torch.get_num_threads())
c = t(lambda i=0: pin.copy_(src[(i % 4) * rows:(i % 4 + 1) * rows]))
print(f"convert f64->f32 pinned, torch copy_: {c*1e3:.1f} ms = {rows*G*4/c/1e9:.1f} GB/s (f32 bytes)")
for nthr in (2, 4, 8, 16):
    pool = cf.ThreadPoolExecutor(nthr)
    def staged(i=0):
        lo = (i % 4) * rows; per = rows // nthr
        fs = [pool.submit(lambda a, b: pin[a - lo:b - lo].copy_(src[a:b]), lo + j * per, lo + (j + 1) * per) for j in range(nthr)]
        [f.result() for f in fs]
    c2 = t(staged)
    print(f"  {nthr} python threads: {c2*1e3:.1f} ms = {rows*G*4/c2/1e9:.1f} GB/s")
torch.set_num_threads(1)
for nthr in (8, 16):
    pool = cf.ThreadPoolExecutor(nthr)
    def staged(i=0):
        lo = (i % 4) * rows; per = rows // nthr
        fs = [pool.submit(lambda a, b: pin[a - lo:b - lo].copy_(src[a:b]), lo + j * per, lo + (j + 1) * per) for j in range(nthr)]
        [f.result() for f in fs]
    c2 = t(staged)
    print(f"  {nthr} python threads, torch threads 1: {c2*1e3:.1f} ms = {rows*G*4/c2/1e9:.1f} GB/s")
h = t(lambda i=0: dev.copy_(pin, non_blocking=True))
print(f"H2D pinned: {h*1e3:.1f} ms = {rows*G*4/h/1e9:.1f} GB/s")
src32 = src.float()
c3 = t(lambda i=0: pin.copy_(src32[(i % 4) * rows:(i % 4 + 1) * rows]))
print(f"copy f32->f32 pinned: {c3*1e3:.1f} ms = {rows*G*4/c3/1e9:.1f} GB/s")
out = torch.empty(rows, 3000, device="cuda"); pout = torch.empty(rows, 3000, pin_memory=True)
d = t(lambda i=0: pout.copy_(out, non_blocking=True))
print(f"D2H pinned: {d*1e3:.2f} ms = {rows*3000*4/d/1e9:.1f} GB/s")
