import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, bench
from onto_vae_amd.vae_model import FusedAdamW
from onto_vae_amd.parallel import dp_train_step
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model, masks = bench.build_model(dev)
X = bench.synth_X(8192, 20000, dev)
opt = FusedAdamW(model, lr=1e-4)
loss = torch.zeros(3, device=dev)
for B in [int(a) for a in sys.argv[1:]] or [128]:
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    for mode in ("0", "1"):
        os.environ["ONTOVAE_DP_ALLREDUCE"] = mode
        for _ in range(20): dp_train_step(model, X, idx, None, 1e-4, opt, loss, reduce_loss=False)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): dp_train_step(model, X, idx, None, 1e-4, opt, loss, reduce_loss=False)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"B {B} allreduce={mode} enqueue us/step {(t1-t0)/n*1e6:.1f} total us/step {(t2-t0)/n*1e6:.1f}", flush=True)
dist.destroy_process_group()
