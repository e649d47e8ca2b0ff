"""What the vendor's fp32 GEMM reaches on this step's two large products (same shapes, same box): a yardstick for
ov_gemm_lds_kernel and ov_dw0_pipe_kernel.  H = 360 hidden units, G = 20000 genes."""
import torch, sys
torch.backends.cuda.matmul.allow_tf32 = False
H, G = 360, 20000
dev = "cuda"
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B in (128, 1024, 2048, 8192):
    x = torch.randn(B, G, device=dev); W0 = torch.randn(H, G, device=dev); dh = torch.randn(H, B, device=dev)
    xt = x.t().contiguous()
    fl = 2.0 * H * G * B
    r = {}
    r["fwd  W0[H,G] @ x^T (x sample-major)"] = t(lambda: torch.matmul(W0, x.t()))
    r["fwd  x @ W0^T"] = t(lambda: torch.matmul(x, W0.t()))
    r["fwd  W0 @ xT[G,B] (feature-major)"] = t(lambda: torch.matmul(W0, xt))
    r["dW0  dh[H,B] @ x[B,G]"] = t(lambda: torch.matmul(dh, x))
    for k, us in r.items():
        print(f"B={B:5d} {k:40s} {us:8.1f} us  {fl/us/1e6:6.1f} TF/s  {fl/us/1e6/157:.2f} of 157", flush=True)
