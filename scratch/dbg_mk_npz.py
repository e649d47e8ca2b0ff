import numpy as np, sys
B = int(sys.argv[1]); seed = int(sys.argv[2])
rng = np.random.RandomState(seed)
idx = rng.randint(0, 2048, size=B).astype(np.int32)
eps = rng.standard_normal((1, B, 360)).astype(np.float32)
np.savez("/tmp/dp_gobp.npz", gobp=1, nu=3, idx=idx, eps=eps, B=B, lr=1e-4, use_idx=1)
