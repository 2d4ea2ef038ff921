import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nenbody_amd as nb
nb.reload_env()  # tools/ read the NB_* kernel-form knobs; a host that merely loads the library does not (nb_diag_enable_env)
def run(n, env, steps=1, z=False):
    for k in ("NB_FAST_SL","NB_FAST_PAIRS","NB_FORCE_3D"): os.environ.pop(k, None)
    os.environ.update(env); nb.reload_env()
    pos, vel = nb.init_state(n, 7)
    if z:
        rng = np.random.default_rng(1); pos[:,2] = rng.uniform(-100,100,n).astype(np.float32)
    with nb.Scene(pos, vel, nb.default_params(mode=nb.NB_MODE_FAST)) as sc:
        sc.step_n(steps); p,v = sc.state()
    return p, v, vel
for n in (4096, 4352, 6144, 8192, 16384, 65536):
    for z in (False, True):
        p0,v0,vel = run(n, {}, z=z)
        p1,v1,_ = run(n, {"NB_FAST_PAIRS":"1"}, z=z)
        dv = np.abs(v0 - vel).max()
        print(n, z, "max|dv|", dv, "max diff", np.abs(v1-v0).max(), "rel", np.abs(v1-v0).max()/dv, "planned", nb._lib.planned_kernels(nb.default_params(mode=nb.NB_MODE_FAST), n, n)[:1], flush=True)
