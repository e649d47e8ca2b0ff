"""Only the Wilcoxon kernel, GO-BP shape (3000 columns x 4096 samples), for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from onto_vae_amd import stats as dstats
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(3000, n, device=dev, generator=g)
b = a + 0.3 * torch.randn(3000, n, device=dev, generator=g)
stat = torch.empty(3000, dtype=torch.float64, device=dev)
p = torch.empty_like(stat)
for _ in range(5):
    dstats.wilcoxon_zsplit(a, b, n, "greater", out=(stat, p))
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10):
    dstats.wilcoxon_zsplit(a, b, n, "greater", out=(stat, p))
ev[1].record()
torch.cuda.synchronize()
print(f"n {n}: {ev[0].elapsed_time(ev[1]) * 100:.1f} us per call")
# the same after 0.3 s of dense work (clock / power state) and on data with a structure closer to activities
w = torch.randn(8192, 8192, device=dev)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for _ in range(40):
    w2 = w @ w
t0.record()
for _ in range(10):
    dstats.wilcoxon_zsplit(a, b, n, "greater", out=(stat, p))
t1.record()
torch.cuda.synchronize()
print(f"after dense work: {t0.elapsed_time(t1) * 100:.1f} us per call")
a2 = torch.rand(3000, 1, device=dev) * 5 + torch.randn(3000, n, device=dev, generator=g) * 0.05
b2 = a2 + torch.randn(3000, n, device=dev, generator=g) * 0.05 + 0.01
for _ in range(3):
    dstats.wilcoxon_zsplit(a2, b2, n, "greater", out=(stat, p))
torch.cuda.synchronize()
t0.record()
for _ in range(10):
    dstats.wilcoxon_zsplit(a2, b2, n, "greater", out=(stat, p))
t1.record()
torch.cuda.synchronize()
d = (a2 - b2).abs()
ties = sum(int(n - torch.unique(d[i]).numel()) for i in range(0, 3000, 300)) / 10
print(f"offset data (values ~ 0..5, differences ~ 0.07): {t0.elapsed_time(t1) * 100:.1f} us per call, tied differences per column ~ {ties:.0f}")
