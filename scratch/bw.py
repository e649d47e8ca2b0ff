import sys; sys.path.insert(0,'.')
import torch, time
import bench
from onto_vae_amd.vae_model import FusedAdamW
dev=torch.device('cuda',0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
a=torch.empty(64*1024*1024,device=dev); b=torch.empty_like(a)
t=timeit(lambda: b.copy_(a)); print('copy 256MB->256MB: %.1f us  %.2f TB/s (r+w)'%(t, 2*a.numel()*4/t/1e6))
a2=torch.empty(7*1024*1024,device=dev); b2=torch.empty_like(a2)
t=timeit(lambda: b2.copy_(a2)); print('copy 28MB: %.1f us  %.2f TB/s (r+w)'%(t, 2*a2.numel()*4/t/1e6))
t=timeit(lambda: a2.mul_(1.0001)); print('inplace scale 28MB: %.1f us  %.2f TB/s (r+w)'%(t, 2*a2.numel()*4/t/1e6))
model,masks=bench.build_model(dev); e=model._engine; opt=FusedAdamW(model,lr=1e-4); e.ensure_grad_buffer()
t=timeit(lambda: e.adamw_apply(opt.next_hyper())); n=e.P.numel(); print('flat adamw %d floats: %.1f us  %.2f TB/s (7 streams)'%(n,t,7*n*4/t/1e6))
