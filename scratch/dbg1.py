import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_gpu_parity import make_model, medium
from ontovae_oracle import OracleOntoVAE
masks, sd = medium()
B=128; rng=np.random.RandomState(B)
o=OracleOntoVAE(masks,3); o.load_state_dict(sd)
model,_=make_model(masks,3,sd)
x=np.log1p(rng.poisson(0.5,size=(B,1000))).astype(np.float32); eps=rng.standard_normal((B,60)).astype(np.float32)
rec,kl,loss,g=o.gradients(x,eps,1e-4)
out=torch.zeros(3,device='cuda')
e=model._engine
e.train_step(torch.as_tensor(x),None,torch.as_tensor(eps),1e-4,None,out,apply_update=False)
gw=e.view('W0',e.Gr).cpu().numpy(); ow=g['encoder.encoder.0.0.weight']
d=np.abs(gw-ow); print('dW0 maxabs',d.max(),'scale',np.abs(ow).max(), 'median |g|', np.median(np.abs(ow)))
bad=np.argwhere(d>1e-4*np.abs(ow).max()); print(len(bad), bad[:20])
rel = d/np.maximum(np.abs(ow),1e-30); print('elements with rel>1e-3:', (rel>1e-3).sum(), 'their |g| max', np.abs(ow)[rel>1e-3].max() if (rel>1e-3).any() else 0)
for n in ['encoder.mu.0.weight','encoder.encoder.0.1.weight','encoder.encoder.0.0.bias']:
    pass
gh=e.view('Wmu',e.Gr).cpu().numpy(); print('dWmu', np.abs(gh-g['encoder.mu.0.weight']).max(), np.abs(g['encoder.mu.0.weight']).max())
print('row-wise max|g| of W0 grad (oracle):', np.sort(np.abs(ow).max(1))[:10])
