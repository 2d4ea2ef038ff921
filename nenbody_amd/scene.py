"""Host-side mirror of the reference's update interface for the n-body path, over the C ABI.

The reference keeps the simulation state in ``main()`` (``positions``, ``velocities``, ``old_positions``,
``old_velocities``, ``instance_data``: src/main.rs:738-750) and advances it with the free function
``update_instance_nbody`` (src/main.rs:404-441), called once per redraw (call-site shape: src/main.rs:925-931).
Its ``src/scene.rs`` is empty (src/scene.rs:1); :class:`Scene` is the type that module was evidently meant
to hold, and :func:`update_instance_nbody` keeps the reference's own five-argument signature.

Everything here is plumbing (numpy arrays in, ctypes calls, numpy arrays out).  The arithmetic runs in
libnenbody_hip.so on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib
from ._lib import NB_MODE_FAST, NB_MODE_STRICT, NbBoidsParams, NbError, NbParams, check  # noqa: F401  (re-exported)

__all__ = ["Scene", "update_instance_nbody", "update_instance_boids", "update_release", "init_state", "NB_MODE_STRICT", "NB_MODE_FAST",
           "NbParams", "NbBoidsParams", "NbError"]


def _as_f32(a, shape_tail, name):
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim != len(shape_tail) + 1 or tuple(arr.shape[1:]) != tuple(shape_tail):
        raise ValueError(f"{name} must have shape (n, {', '.join(map(str, shape_tail))}), got {arr.shape}")
    return arr


def camera_constant(vertical_fov_deg: float, aspect_ratio: float, near: float = 1.0, far: float = 10000.0) -> np.ndarray:
    """The constant of a camera array, OPENGL_TO_WGPU_MATRIX * cgmath::perspective(...) (src/gfx.rs:12-17, 365, 367), as the
    (4, 4) array ``Scene.cameras`` takes ([k] = column k).  The reference's CameraArray::new derives the vertical field of view
    as horizontal_fov / aspect_ratio (src/gfx.rs:381) and build_camera passes near = 1, far = 10000."""
    cp = np.zeros((4, 4), np.float32)
    check(_lib.load().nb_camera_constant(vertical_fov_deg, aspect_ratio, near, far, cp.ctypes.data))
    return cp


def init_state(n: int, seed: int = 1234):
    """Seeded stand-in for the reference's unseeded initial state (src/main.rs:736-747).

    Returns (positions, velocities), float32 arrays of shape (n, 3): velocities (U[0,0.1), U[0,0.1), 0)
    drawn first for every body, then positions (U[-100,100), U[-100,100), 0) -- the reference's
    distributions and draw order.
    """
    pos = np.empty((n, 3), np.float32)
    vel = np.empty((n, 3), np.float32)
    check(_lib.load().nb_init_state(seed, n, pos.ctypes.data, vel.ctypes.data))
    return pos, vel


class Scene:
    """Device-resident n-body state with the update of src/main.rs:404-441 as :meth:`step`.

    Host mirrors (``positions``, ``velocities``, ``instances``) are refreshed by :meth:`step`, so the
    consumers at src/main.rs:932-945 (instance upload, cameras) would read them unchanged;
    :meth:`step_n` keeps everything on the device and is the benchmark path.
    """

    def __init__(self, positions, velocities, params: Optional[NbParams] = None):
        lib = _lib.load()
        pos = _as_f32(positions, (3,), "positions")
        vel = _as_f32(velocities, (3,), "velocities")
        if len(pos) != len(vel):
            raise ValueError("positions and velocities must have the same length")
        if len(pos) == 0:
            raise ValueError("a Scene needs at least one body")
        self.n = len(pos)
        self.params = params if params is not None else _lib.default_params()
        self._lib = lib
        self._ctx = ctypes.c_void_p()
        check(lib.nb_create(self.n, 1, ctypes.byref(self.params), ctypes.byref(self._ctx)))
        self._positions = pos.copy()
        self._velocities = vel.copy()
        self._instances = np.zeros((self.n, 4, 4), np.float32)
        try:
            check(lib.nb_upload(self._ctx, self._positions.ctypes.data, self._velocities.ctypes.data), self._ctx)
        except Exception:
            self.close()
            raise

    # -- constructors mirroring the Rust shim ---------------------------------------
    @classmethod
    def new(cls, n: int, params: Optional[NbParams] = None, seed: int = 1234) -> "Scene":
        pos, vel = init_state(n, seed)
        return cls(pos, vel, params)

    @classmethod
    def from_state(cls, positions, velocities, params: Optional[NbParams] = None) -> "Scene":
        return cls(positions, velocities, params)

    # -- stepping ---------------------------------------------------------------------------------------
    def step(self) -> None:
        """One update_instance_nbody, then refresh the host mirrors (positions, velocities, instances)."""
        check(self._lib.nb_step(self._ctx, 1), self._ctx)
        self._refresh(True)

    def step_n(self, k: int) -> None:
        """k updates, device-resident and asynchronous; host mirrors are NOT refreshed (see :meth:`sync`)."""
        check(self._lib.nb_step(self._ctx, int(k)), self._ctx)

    def step_boids(self, params: Optional[NbBoidsParams] = None) -> None:
        """One update_instance_boids (src/main.rs:443-526), then refresh the host mirrors."""
        check(self._lib.nb_step_boids(self._ctx, 1, ctypes.byref(params) if params is not None else None), self._ctx)
        self._refresh(True)

    def step_boids_n(self, k: int, params: Optional[NbBoidsParams] = None) -> None:
        """k boids updates, device-resident and asynchronous; host mirrors are NOT refreshed."""
        check(self._lib.nb_step_boids(self._ctx, int(k), ctypes.byref(params) if params is not None else None), self._ctx)

    def step_random(self, seed: int = 0, k: int = 1) -> None:
        """k random-walk updates (update_instance_random, src/main.rs:381-402), then refresh the host mirrors."""
        check(self._lib.nb_step_random(self._ctx, int(k), int(seed)), self._ctx)
        self._refresh(True)

    def cameras(self, up, cp) -> np.ndarray:
        """CameraArray::update (src/gfx.rs:397-408) for the current state: per body cp * look_at_dir(position, velocity,
        up).  ``cp`` is the array's constant correction*proj as a (4, 4) array whose [k] is column k.  Returns (n, 4, 4)."""
        upv = np.ascontiguousarray(up, np.float32).reshape(3)
        cpm = np.ascontiguousarray(cp, np.float32).reshape(16)
        out = np.zeros((self.n, 4, 4), np.float32)
        check(self._lib.nb_cameras(self._ctx, upv.ctypes.data, cpm.ctypes.data, out.ctypes.data), self._ctx)
        return out

    def device_state(self, with_instances: bool = True):
        """Device pointers (ints) of the current position records, velocity records and model matrices: the zero-copy
        hand-off.  Valid until the next step / upload / close."""
        p, v, m = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check(self._lib.nb_device_state(self._ctx, ctypes.byref(p), ctypes.byref(v), ctypes.byref(m) if with_instances else None),
              self._ctx)
        return p.value, v.value, (m.value if with_instances else None)

    def sync(self) -> None:
        check(self._lib.nb_sync(self._ctx), self._ctx)

    @property
    def steps_done(self) -> int:
        return int(self._lib.nb_steps_done(self._ctx))

    # -- state access -----------------------------------------------------------------------------------
    def _refresh(self, with_instances: bool) -> None:
        inst = self._instances.ctypes.data if with_instances else None
        check(self._lib.nb_download(self._ctx, self._positions.ctypes.data, self._velocities.ctypes.data, inst), self._ctx)

    def positions(self) -> np.ndarray:
        self._refresh(False)
        return self._positions

    def velocities(self) -> np.ndarray:
        self._refresh(False)
        return self._velocities

    def instances(self) -> np.ndarray:
        """Model matrices of the current state, shape (n, 4, 4); [k] is column k (column-major, main.rs:437-439)."""
        self._refresh(True)
        return self._instances

    def state(self):
        self._refresh(False)
        return self._positions.copy(), self._velocities.copy()

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.nb_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _check_update_args(instances, positions, old_positions, velocities, old_velocities):
    for name, arr, tail in (("instances", instances, (4, 4)), ("positions", positions, (3,)),
                            ("old_positions", old_positions, (3,)), ("velocities", velocities, (3,)),
                            ("old_velocities", old_velocities, (3,))):
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.flags.c_contiguous and arr.flags.writeable):
            raise TypeError(f"{name} must be a writable C-contiguous float32 numpy array")
        if arr.ndim != len(tail) + 1 or tuple(arr.shape[1:]) != tail:
            raise ValueError(f"{name} must have shape (n, {', '.join(map(str, tail))})")


def _update_call(fn, instances, positions, old_positions, velocities, old_velocities, params):
    _check_update_args(instances, positions, old_positions, velocities, old_velocities)
    args = []
    for arr in (instances, positions, old_positions, velocities, old_velocities):
        args += [arr.ctypes.data if len(arr) else None, len(arr)]
    rc = fn(*args, ctypes.byref(params) if params is not None else None)
    if rc == _lib.NB_ERR_INVALID:  # the reference panics here (copy_from_slice, indexing); in Python that is a ValueError
        raise ValueError(_lib.load().nb_last_error(None).decode())
    check(rc)


def update_instance_boids(instances, positions, old_positions, velocities, old_velocities,
                          params: Optional[NbBoidsParams] = None) -> None:
    """The reference's live controller with its own five arguments, updated in place (src/main.rs:443-449).

    Same contract as :func:`update_instance_nbody` (snapshot copies first, main.rs:459-460; `zip` stops at the
    shortest of instances / positions / velocities, main.rs:465-469).  The position folds run over all of
    ``old_positions`` and the velocity fold (main.rs:494-504) over all of ``old_velocities``, each with its own length.
    One call of ``nb_update_instance_boids`` (include/nenbody.h).
    """
    _update_call(_lib.load().nb_update_instance_boids, instances, positions, old_positions, velocities, old_velocities, params)


def update_instance_nbody(instances, positions, old_positions, velocities, old_velocities,
                          params: Optional[NbParams] = None) -> None:
    """The reference's operator, same five arguments, updated in place (src/main.rs:404-410).

    ``instances`` is (n, 4, 4) float32, the others (n, 3) float32 numpy arrays (they must be writable,
    C-contiguous float32: they are the caller's ``Vec``s).  Behaviour kept from the reference:
      * ``old_positions`` / ``old_velocities`` receive copies of the inputs (main.rs:415-416); a length
        mismatch is an error, as ``copy_from_slice`` panics;
      * ``instances.zip(positions).zip(velocities)`` stops at the shortest of the three (main.rs:420-423):
        only that many bodies are updated, while the fold still runs over all of ``old_positions``.
    One call of ``nb_update_instance_nbody`` (include/nenbody.h): one upload, one step, one download; the device context
    is kept inside the library between calls.  A caller that steps repeatedly without reading the state back should
    hold a :class:`Scene` instead.
    """
    _update_call(_lib.load().nb_update_instance_nbody, instances, positions, old_positions, velocities, old_velocities, params)


def _random_args(instances, positions, velocities):
    for name, arr, tail in (("instances", instances, (4, 4)), ("positions", positions, (3,)), ("velocities", velocities, (3,))):
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.flags.c_contiguous and arr.flags.writeable):
            raise TypeError(f"{name} must be a writable C-contiguous float32 numpy array")
        if arr.ndim != len(tail) + 1 or tuple(arr.shape[1:]) != tail:
            raise ValueError(f"{name} must have shape (n, {', '.join(map(str, tail))})")
    args = []
    for arr in (instances, positions, velocities):
        args += [arr.ctypes.data if len(arr) else None, len(arr)]
    return args


def update_instance_random(instances, positions, velocities) -> None:
    """The reference's third controller with its own three arguments, updated in place (src/main.rs:381-385): every body's
    velocity takes a small random kick in x and y, the position follows, the matrix is rebuilt; `zip` stops at the shortest of
    the three (main.rs:386-389).  The reference draws from an unseeded ``thread_rng``; here body n at the k-th call draws from
    the counter-based stream (seed, k, n), seed and call counter kept by the library (:func:`update_random_seed` sets the
    seed and restarts the counter), so a run is reproducible.  One call of ``nb_update_instance_random``."""
    check(_lib.load().nb_update_instance_random(*_random_args(instances, positions, velocities)))


def update_instance_random_seeded(instances, positions, velocities, seed: int = 0, step: int = 0) -> None:
    """:func:`update_instance_random` with the stream position given by the caller: body n draws from (seed, step, n).
    One call of ``nb_update_instance_random_seeded``."""
    check(_lib.load().nb_update_instance_random_seeded(*_random_args(instances, positions, velocities), int(seed), int(step)))


def update_random_seed(seed: int) -> None:
    """Seed of :func:`update_instance_random`'s stream; restarts its call counter."""
    _lib.load().nb_update_random_seed(int(seed))


def update_release() -> None:
    """Frees the device contexts the drop-in functions keep between calls."""
    _lib.load().nb_update_release()
