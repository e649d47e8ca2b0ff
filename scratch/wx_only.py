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
