import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, nenbody_amd as nb
from nenbody_amd.dist import HipBackend
be = HipBackend(); dev = torch.device("cuda", 0)
n_total = 131072
pos, vel = nb.init_state(n_total, 1234)
cur = torch.zeros((n_total, 4)); cur[:, :3] = torch.from_numpy(pos); cur = cur.to(dev); nxt = torch.zeros_like(cur)
for env in ({"NB_STRICT_PC":1}, {"NB_STRICT_PC":1,"NB_STRICT_FORCE_IEEE":1}, {"NB_STRICT_PC":1,"NB_FORCE_3D":1}, {"NB_STRICT_PC":0,"NB_STRICT_LANES":1}, {"NB_STRICT_PC":0,"NB_STRICT_LANES":1,"NB_STRICT_FORCE_IEEE":1}):
    for k,v in env.items(): os.environ[k]=str(v)
    for count in (16384, 8192, 4096):
        params = nb.default_params(); v4 = torch.zeros((count,4), device=dev)
        for _ in range(2): be.step(params, n_total, 0, count, cur, nxt, v4, None)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(4): be.step(params, n_total, 0, count, cur, nxt, v4, None)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/4
        print(env, count, f"{dt*1e3:.3f} ms", flush=True)
    for k in env: os.environ.pop(k)
