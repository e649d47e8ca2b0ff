import time, torch
torch.cuda.init(); torch.empty(1, device="cuda")
for shape in ((4096, 20000), (4096, 20000), (32768, 3000), (32768, 3000)):
    t0 = time.perf_counter(); a = torch.empty(*shape, dtype=torch.float32, pin_memory=True); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); a.zero_(); dz = time.perf_counter() - t0
    print(shape, f"pinned alloc {dt*1e3:.1f} ms, first touch {dz*1e3:.1f} ms")
    del a
t0 = time.perf_counter(); a = torch.empty(32768, 3000); a.zero_(); print(f"pageable alloc+touch {1e3*(time.perf_counter()-t0):.1f} ms")
