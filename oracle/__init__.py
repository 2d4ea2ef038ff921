"""ctypes wrapper of the CPU oracle (oracle/nbody_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
nenbody_amd never imports this package.  Parity of the oracle itself is UNPINNED by the reference (it ships no
tests and cannot be built here); see the header of nbody_oracle.c.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnbody_oracle.so")

# reference constants, src/main.rs:411-413
DT, G, BIAS = np.float32(0.1), np.float32(0.001), np.float32(0.0000001)

_lib = None


class BoidsParams(ctypes.Structure):
    """nbo_boids_params: the constants of update_instance_boids, src/main.rs:450-456."""

    _fields_ = [(k, ctypes.c_float) for k in ("dt", "rule_1_distance", "rule_2_distance", "rule_3_distance", "rule_1_scale",
                                               "rule_2_scale", "rule_3_scale")]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libnbody_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = ctypes.CDLL(LIB_PATH)
        vp, u32, u64, f, i = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_float, ctypes.c_int
        lib.nbo_init_state.argtypes = [u64, u32, vp, vp]
        lib.nbo_init_state.restype = None
        lib.nbo_instances.argtypes = [vp, vp, vp, u32]
        lib.nbo_instances.restype = None
        lib.nbo_step_range.argtypes = [vp, vp, vp, vp, u32, u32, u32, f, f, f]
        lib.nbo_step_range.restype = None
        lib.nbo_run.argtypes = [vp, vp, vp, u32, u32, f, f, f, i]
        lib.nbo_run.restype = i
        lib.nbo_run_batched.argtypes = [vp, vp, vp, u32, u32, f, f, f, i]
        lib.nbo_run_batched.restype = i
        lib.nbo_step_range_batched.argtypes = [vp, vp, vp, vp, u32, u32, u32, f, f, f]
        lib.nbo_step_range_batched.restype = None
        lib.nbo_batched_available.restype = i
        lib.nbo_run_f64.argtypes = [vp, vp, u32, u32, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        lib.nbo_run_f64.restype = i
        lib.nbo_step_range_dv_f64.argtypes = [vp, vp, u32, u32, u32, ctypes.c_double, ctypes.c_double, ctypes.c_double]
        lib.nbo_step_range_dv_f64.restype = None
        lib.nbo_cameras.argtypes = [vp, vp, vp, vp, vp, u32]
        lib.nbo_cameras.restype = None
        lib.nbo_random_step_range.argtypes = [vp, vp, vp, u32, u32, u64, u64]
        lib.nbo_random_step_range.restype = None
        lib.nbo_boids_default_params.argtypes = [ctypes.POINTER(BoidsParams)]
        lib.nbo_boids_default_params.restype = None
        lib.nbo_boids_step_range.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, ctypes.POINTER(BoidsParams)]
        lib.nbo_boids_step_range.restype = None
        lib.nbo_boids_step_range2.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, u32, ctypes.POINTER(BoidsParams)]
        lib.nbo_boids_step_range2.restype = None
        lib.nbo_boids_run.argtypes = [vp, vp, vp, u32, u32, ctypes.POINTER(BoidsParams), i]
        lib.nbo_boids_run.restype = i
        _lib = lib
    return _lib


def cpu_limits() -> dict:
    """What this process may actually use: the affinity mask AND the cgroup CPU quota (a container is commonly given every
    logical CPU in its mask but a quota of a few cores' worth of time: threads beyond the quota only get throttled).
    `threads` = the worker count to use: min(affinity, ceil(quota))."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        affinity = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except Exception:
            pass
    threads = affinity if quota is None else max(1, min(affinity, int(-(-quota // 1))))
    return {"affinity": affinity, "cgroup_quota_cores": quota, "threads": threads}


def throttled_usec() -> int:
    """cgroup v2 cpu.stat throttled_usec (0 if unreadable): a run during which this grows was capped by the quota."""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1])
    except Exception:
        pass
    return 0


def ncores() -> int:
    """Worker threads the oracle uses by default: the cores this process can really run on (cpu_limits)."""
    return cpu_limits()["threads"]


def init_state(n: int, seed: int = 1234):
    pos = np.empty((n, 3), np.float32)
    vel = np.empty((n, 3), np.float32)
    load().nbo_init_state(seed, n, pos.ctypes.data, vel.ctypes.data)
    return pos, vel


def run(pos, vel, k: int, dt=DT, g=G, bias=BIAS, threads: int = 0, want_instances: bool = False, batched: bool = False):
    """k applications of update_instance_nbody (main.rs:404-441).  Returns new (pos, vel[, instances]).

    batched=True runs eight bodies per AVX2 vector (nbo_step_range_batched: the same scalar operations per body, the same
    bits, several times faster); the default is the scalar loop."""
    p = np.ascontiguousarray(pos, np.float32).copy()
    v = np.ascontiguousarray(vel, np.float32).copy()
    n = len(p)
    inst = np.zeros((n, 4, 4), np.float32) if want_instances else None
    fn = load().nbo_run_batched if batched else load().nbo_run
    rc = fn(p.ctypes.data, v.ctypes.data, inst.ctypes.data if want_instances else None, n, k, dt, g, bias,
            threads or ncores())
    assert rc == 0
    return (p, v, inst) if want_instances else (p, v)


def step_range(old_pos, vel_range, first: int, count: int, dt=DT, g=G, bias=BIAS, want_instances: bool = False):
    """One step for bodies [first, first+count) against the snapshot of all positions (one thread)."""
    old = np.ascontiguousarray(old_pos, np.float32)
    v = np.ascontiguousarray(vel_range, np.float32).copy()
    assert len(v) == count
    p = np.empty((count, 3), np.float32)
    inst = np.zeros((count, 4, 4), np.float32) if want_instances else None
    load().nbo_step_range(old.ctypes.data, p.ctypes.data, v.ctypes.data, inst.ctypes.data if want_instances else None,
                          len(old), first, count, dt, g, bias)
    return (p, v, inst) if want_instances else (p, v)


def instances(pos, vel):
    p = np.ascontiguousarray(pos, np.float32)
    v = np.ascontiguousarray(vel, np.float32)
    inst = np.zeros((len(p), 4, 4), np.float32)
    load().nbo_instances(p.ctypes.data, v.ctypes.data, inst.ctypes.data, len(p))
    return inst


def run_f64(pos, vel, k: int, dt=0.1, g=0.001, bias=0.0000001):
    p = np.ascontiguousarray(pos, np.float64).copy()
    v = np.ascontiguousarray(vel, np.float64).copy()
    rc = load().nbo_run_f64(p.ctypes.data, v.ctypes.data, len(p), k, dt, g, bias)
    assert rc == 0
    return p, v


def step_range_dv_f64(old_pos, first: int, count: int, dt=0.1, g=0.001, bias=0.0000001):
    """One step's velocity change of bodies [first, first+count), accumulated in binary64 from the binary32 snapshot:
    the yardstick for the rounding error of a binary32 sum (the reference's included).  Note the binary64 constants are
    the decimal literals of main.rs:411-413, not the binary32 roundings of them: callers compare at tolerances far above
    that difference (1e-8 relative)."""
    old = np.ascontiguousarray(old_pos, np.float32)
    dv = np.empty((count, 3), np.float64)
    load().nbo_step_range_dv_f64(old.ctypes.data, dv.ctypes.data, len(old), first, count, float(dt), float(g), float(bias))
    return dv


def boids_params() -> BoidsParams:
    p = BoidsParams()
    load().nbo_boids_default_params(ctypes.byref(p))
    return p


def boids_run(pos, vel, k: int, params: BoidsParams = None, threads: int = 0, want_instances: bool = False):
    """k applications of update_instance_boids (main.rs:443-526).  Returns new (pos, vel[, instances])."""
    p = np.ascontiguousarray(pos, np.float32).copy()
    v = np.ascontiguousarray(vel, np.float32).copy()
    n = len(p)
    bp = params if params is not None else boids_params()
    inst = np.zeros((n, 4, 4), np.float32) if want_instances else None
    rc = load().nbo_boids_run(p.ctypes.data, v.ctypes.data, inst.ctypes.data if want_instances else None, n, k,
                              ctypes.byref(bp), threads or ncores())
    assert rc == 0
    return (p, v, inst) if want_instances else (p, v)


def boids_step_range(old_pos, old_vel, first: int, count: int, params: BoidsParams = None, want_instances: bool = False):
    """One boids step for bodies [first, first+count) against the snapshots of all positions and velocities."""
    op = np.ascontiguousarray(old_pos, np.float32)
    ov = np.ascontiguousarray(old_vel, np.float32)
    bp = params if params is not None else boids_params()
    v = ov[first:first + count].copy()
    p = np.empty((count, 3), np.float32)
    inst = np.zeros((count, 4, 4), np.float32) if want_instances else None
    load().nbo_boids_step_range(op.ctypes.data, ov.ctypes.data, p.ctypes.data, v.ctypes.data,
                                inst.ctypes.data if want_instances else None, len(op), first, count, ctypes.byref(bp))
    return (p, v, inst) if want_instances else (p, v)


def boids_update_instance(instances_len: int, positions, velocities, params: BoidsParams = None):
    """update_instance_boids (main.rs:443-526) as the reference's free function behaves on slices of UNEQUAL length:
    the position folds run over all of old_positions, the velocity fold over all of old_velocities, and the zip updates the
    first min(len(instances), len(positions), len(velocities)) bodies.  Returns (new positions, new velocities, instances) of
    that many bodies."""
    op = np.ascontiguousarray(positions, np.float32)
    ov = np.ascontiguousarray(velocities, np.float32)
    bp = params if params is not None else boids_params()
    count = min(int(instances_len), len(op), len(ov))
    v = ov[:count].copy()
    p = np.empty((count, 3), np.float32)
    inst = np.zeros((count, 4, 4), np.float32)
    if count:
        load().nbo_boids_step_range2(op.ctypes.data, len(op), ov.ctypes.data, len(ov), p.ctypes.data, v.ctypes.data, inst.ctypes.data,
                                     0, count, ctypes.byref(bp))
    return p, v, inst


def cameras(eyes, dirs, up, cp):
    """CameraArray::update (gfx.rs:397-408): per entity (correction*proj) * look_at_dir(eye, dir, up); cp is the constant
    correction*proj as a (4, 4) array whose [k] is column k."""
    e = np.ascontiguousarray(eyes, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    u = np.ascontiguousarray(up, np.float32)
    c = np.ascontiguousarray(cp, np.float32)
    out = np.zeros((len(e), 4, 4), np.float32)
    load().nbo_cameras(e.ctypes.data, d.ctypes.data, u.ctypes.data, c.ctypes.data, out.ctypes.data, len(e))
    return out


def camera_constant(vertical_fov_deg: float, aspect: float, near: float = 1.0, far: float = 10000.0):
    """OPENGL_TO_WGPU_MATRIX * cgmath::perspective(...) of build_camera (gfx.rs:365-367) as a (4, 4) array whose [k] is
    column k; raises ValueError where cgmath would panic."""
    cp = np.zeros((4, 4), np.float32)
    lib = load()
    lib.nbo_camera_constant.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    lib.nbo_camera_constant.restype = ctypes.c_int
    if lib.nbo_camera_constant(vertical_fov_deg, aspect, near, far, cp.ctypes.data) != 0:
        raise ValueError("perspective: arguments cgmath asserts against")
    return cp


def random_run(pos, vel, k: int, seed: int, first_step: int = 0, want_instances: bool = False):
    """k applications of update_instance_random (main.rs:381-402) with the build-owned counter-based stream."""
    p = np.ascontiguousarray(pos, np.float32).copy()
    v = np.ascontiguousarray(vel, np.float32).copy()
    inst = np.zeros((len(p), 4, 4), np.float32) if want_instances else None
    for s in range(k):
        last = want_instances and s + 1 == k
        load().nbo_random_step_range(p.ctypes.data, v.ctypes.data, inst.ctypes.data if last else None, 0, len(p), seed,
                                     first_step + s)
    return (p, v, inst) if want_instances else (p, v)
