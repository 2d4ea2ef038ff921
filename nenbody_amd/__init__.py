"""nenbody_amd -- MI355X-native implementation of nenbody's all-pairs gravity + Euler step.

One path only: ``update_instance_nbody`` (reference src/main.rs:404-441), behind a C ABI
(include/nenbody.h, libnenbody_hip.so).  This package is the thin host side: a ctypes binding, a
``Scene`` that mirrors the reference's update interface, and a sharded scene for one process per GPU.
"""
from ._lib import (NB_MODE_FAST, NB_MODE_STRICT, NbBoidsParams, NbError, NbParams, default_boids_params,  # noqa: F401
                   default_params, load)
from .scene import (Scene, camera_constant, init_state, update_instance_boids, update_instance_nbody,  # noqa: F401
                    update_instance_random, update_instance_random_seeded, update_random_seed, update_release)
from .dist import NativeShard, ShardedScene, comm_id, partition  # noqa: F401



def reload_env() -> None:
    """Have the library read its NB_* diagnostic overrides again (they are read once per process; tools that change
    os.environ between runs call this: ``nb_debug_reload_env``)."""
    load().nb_debug_reload_env()


__all__ = ["reload_env", "Scene", "ShardedScene", "NativeShard", "comm_id", "partition", "init_state", "camera_constant", "update_instance_nbody", "update_instance_boids", "update_instance_random", "update_instance_random_seeded", "update_random_seed", "update_release",
           "default_params", "default_boids_params", "load", "NbParams", "NbBoidsParams", "NbError", "NB_MODE_STRICT",
           "NB_MODE_FAST"]
