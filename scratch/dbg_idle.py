import torch, time, sys
x = torch.zeros(1 << 28, device="cuda")
y = (x + 1).sum().item()
print("idle ready", y, flush=True)
time.sleep(float(sys.argv[1]))
