"""Body set sharded over one process per GPU, one all-gather of positions per step.

Within a step every body depends only on the start-of-step snapshot of ALL positions
(``old_positions``, src/main.rs:415, 425), so the set shards by contiguous index range with exactly one
exchange: after the local update each rank contributes its new positions and receives everyone else's
(RCCL all-gather over xGMI; ``torch.distributed`` backend "nccl" is RCCL on ROCm).  Velocities never move.
STRICT results do not depend on the world size: a body's fold order over j is unchanged by sharding.

torch is plumbing here: device buffers, the stream, the collective.  The step itself is
``nb_launch_step`` (include/nenbody.h) on the raw device pointers.
"""
from __future__ import annotations

import ctypes
import time
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import NbParams, check

__all__ = ["partition", "ShardedScene", "HipBackend", "NativeShard", "comm_id"]


def partition(n: int, world: int) -> List[Tuple[int, int]]:
    """(first, count) of every rank: equal slots of ceil(n/world) bodies; trailing slots may be short or empty.

    Equal slots keep the all-gather a plain (non-v) collective: the position buffer holds
    ``world * slot`` records, of which the kernel only ever reads the first n.
    """
    if n <= 0 or world <= 0:
        raise ValueError("partition needs n > 0 and world > 0")
    slot = -(-n // world)
    out = []
    for r in range(world):
        first = min(r * slot, n)
        out.append((first, max(0, min(n - first, slot))))
    return out


class HipBackend:
    """The product compute backend: libnenbody_hip.so's launch API on torch-owned device memory."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load()

    def scratch_bytes(self, params: NbParams, n_total: int, count: int) -> int:
        return int(self.lib.nb_scratch_bytes(ctypes.byref(params), n_total, count))

    def step(self, params, n_total, first, count, pos_in, pos_out, vel, scratch) -> None:
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        sp = scratch.data_ptr() if scratch is not None and scratch.numel() else None
        sb = scratch.numel() if scratch is not None else 0
        with torch.cuda.device(pos_in.device):   # the launch and the device status word belong to the buffers' device
            check(self.lib.nb_launch_step(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), pos_out.data_ptr(),
                                          vel.data_ptr(), sp, sb, stream))

    def scratch_bytes_phased(self, params: NbParams, n_total: int, count: int, j_lo: int, j_hi: int) -> int:
        return int(self.lib.nb_scratch_bytes_phased(ctypes.byref(params), n_total, count, j_lo, j_hi))

    def step_phase(self, params, n_total, first, count, j_lo, j_hi, phase, pos_in, pos_out, vel, scratch) -> None:
        """FAST only: ``nb_launch_step_phase`` -- phase 0 folds records [j_lo, j_hi), phase 1 the rest and integrates."""
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        with torch.cuda.device(pos_in.device):
            check(self.lib.nb_launch_step_phase(ctypes.byref(params), n_total, first, count, j_lo, j_hi, phase, pos_in.data_ptr(),
                                                pos_out.data_ptr(), vel.data_ptr(), scratch.data_ptr(), scratch.numel(), stream))

    # -- FAST on shards, every unordered pair once ("half shell": nb_launch_ring_fold / _finish, include/nenbody.h) ----------
    def ring_partners(self, params: NbParams, n_total: int, first: int, count: int) -> int:
        """D >= 1: the ranks in front of this one that own bodies of its pair lists; 0: this shape keeps ``step``."""
        d = int(self.lib.nb_ring_partners(ctypes.byref(params), n_total, first, count))
        if d < 0:
            check(d)
        return d

    def ring_scratch_bytes(self, params: NbParams, n_total: int, first: int, count: int) -> int:
        return int(self.lib.nb_ring_scratch_bytes(ctypes.byref(params), n_total, first, count))

    def ring_fold(self, params, n_total, first, count, pos_in, sums, scratch) -> None:
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        with torch.cuda.device(pos_in.device):
            check(self.lib.nb_launch_ring_fold(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), sums.data_ptr(),
                                               scratch.data_ptr(), scratch.numel(), stream))

    def ring_phased(self, params: NbParams, n_total: int, first: int, count: int) -> bool:
        """can this shape run its step in phases (``nb_launch_ring_fold_phase``: both exchanges behind compute)?"""
        r = int(self.lib.nb_ring_phased(ctypes.byref(params), n_total, first, count))
        if r < 0:
            check(r)
        return r == 1

    def ring_fold_phase(self, params, n_total, first, count, phase, pos_in, sums, scratch) -> None:
        """one phase of the fold: NB_RING_OWN reads only the rank's own slot of pos_in, NB_RING_REST the whole snapshot (and leaves
        the sums of the ranks in front final), NB_RING_SUMS makes the rank's own sums"""
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        with torch.cuda.device(pos_in.device):
            check(self.lib.nb_launch_ring_fold_phase(ctypes.byref(params), n_total, first, count, int(phase), pos_in.data_ptr(),
                                                     sums.data_ptr(), scratch.data_ptr(), scratch.numel(), stream))

    def ring_finish(self, params, n_total, first, count, pos_in, pos_out, vel, sums, recv) -> None:
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        with torch.cuda.device(pos_in.device):
            check(self.lib.nb_launch_ring_finish(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), pos_out.data_ptr(),
                                                 vel.data_ptr(), sums.data_ptr(), recv.data_ptr(), stream))

    def ring_finish_phase(self, params, n_total, first, count, pos_in, pos_out, vel, sums, recv, scratch) -> None:
        """the finish of a step in phases, fused (``nb_launch_ring_finish_phase``): with ``sums`` None it adds the rank's own
        records itself (NB_RING_SUMS is not launched); it leaves the planes of the new own slot in ``scratch``, so the next step
        starts with NB_RING_OWN_READY"""
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        with torch.cuda.device(pos_in.device):
            check(self.lib.nb_launch_ring_finish_phase(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), pos_out.data_ptr(),
                                                       vel.data_ptr(), sums.data_ptr() if sums is not None else None, recv.data_ptr(),
                                                       scratch.data_ptr(), scratch.numel(), stream))

    def instances(self, count, pos, vel, inst) -> None:
        import torch

        stream = torch.cuda.current_stream(pos.device).cuda_stream
        check(self.lib.nb_launch_instances(count, pos.data_ptr(), vel.data_ptr(), inst.data_ptr(), stream))

    def boids_step(self, params, n_total, first, count, pos_in, vel_in, pos_out, vel_out) -> None:
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        check(self.lib.nb_launch_boids_step(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), vel_in.data_ptr(),
                                            pos_out.data_ptr(), vel_out.data_ptr(), stream))


    def boids_split_scratch_bytes(self, params, n_total, count) -> int:
        return int(self.lib.nb_boids_split_scratch_bytes(ctypes.byref(params), n_total, count))

    def boids_step_split(self, params, n_total, first, count, pos_in, vel_in, pos_out, vel_out, scratch) -> None:
        """the boids step with the j range in slices (``nb_launch_boids_step_split``): the reference's neighbour sets and counts,
        reassociated sums -- the form that lets a small shard fill the chip"""
        import torch

        stream = torch.cuda.current_stream(pos_in.device).cuda_stream
        check(self.lib.nb_launch_boids_step_split(ctypes.byref(params), n_total, first, count, pos_in.data_ptr(), vel_in.data_ptr(),
                                                  pos_out.data_ptr(), vel_out.data_ptr(), scratch.data_ptr(), scratch.numel(), stream))


class ShardedScene:
    """One rank's share of a Scene: bodies [first, first+count) plus a replica of all positions.

    ``positions`` / ``velocities`` are the FULL (n, 3) host arrays, identical on every rank (each rank keeps
    its own slice of the velocities).  ``group`` is a torch.distributed process group (None = default group;
    a world of 1 needs no initialised process group at all).
    """

    def __init__(self, positions, velocities, params: Optional[NbParams] = None, *, device=None, group=None,
                 backend=None, rank: Optional[int] = None, world: Optional[int] = None, overlap: bool = False,
                 ring: Optional[bool] = None, ring_overlap: Optional[bool] = None, exchange: str = "collective"):
        import torch
        import torch.distributed as dist

        self.torch = torch
        self.dist = dist
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if world > 1 else 0
        self.world, self.rank = world, rank
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        vel = np.ascontiguousarray(velocities, dtype=np.float32)
        if pos.ndim != 2 or pos.shape[1] != 3 or pos.shape != vel.shape:
            raise ValueError("positions and velocities must both have shape (n, 3)")
        self.n = len(pos)
        self.params = params if params is not None else _lib.default_params()
        self.backend = backend if backend is not None else HipBackend()
        if device is None:
            if self.backend.name == "hip":
                if not torch.cuda.is_available():
                    raise _lib.NbError(_lib.NB_ERR_NO_DEVICE, "no HIP device visible to torch; nenbody_amd has no CPU path")
                device = torch.device("cuda", torch.cuda.current_device())
            else:
                device = torch.device("cpu")
        self.device = torch.device(device)
        parts = partition(self.n, world)
        self.first, self.count = parts[rank]
        self.slot = -(-self.n // world)
        padded = self.slot * world
        rec = torch.zeros((padded, 4), dtype=torch.float32)
        rec[: self.n, :3] = torch.from_numpy(pos)
        self.pos = [rec.to(self.device), torch.zeros((padded, 4), dtype=torch.float32, device=self.device)]
        vrec = torch.zeros((max(self.count, 1), 4), dtype=torch.float32)
        if self.count:
            vrec[: self.count, :3] = torch.from_numpy(vel[self.first:self.first + self.count])
        self.vel = vrec.to(self.device)
        # FAST, equal ranks: every unordered pair once, the other ranks' halves leaving in a SECOND exchange per step
        # (ring=None: where the library plans it; True: required; False: the ordered fold with its one exchange)
        self.partners = 0
        if ring is not False and world > 1 and self.params.mode == _lib.NB_MODE_FAST and self.n == self.slot * world \
                and hasattr(self.backend, "ring_partners"):
            self.partners = self.backend.ring_partners(self.params, self.n, self.first, self.count)
        if ring and not self.partners:
            raise ValueError("ring=True: this shape does not take the pairs form on shards (FAST, equal ranks of whole blocks)")
        self.sums = self.recv = None
        if self.partners:
            self.sums = torch.zeros(((self.partners + 1) * self.count, 4), dtype=torch.float32, device=self.device)
            self.recv = torch.zeros((self.partners * self.count, 4), dtype=torch.float32, device=self.device)
        self.overlap = bool(overlap) and self.params.mode == _lib.NB_MODE_FAST and world > 1 and not self.partners
        self._pending = None     # the exchange in flight (overlap): a torch.distributed work handle, or None
        self._own_ready = None   # the scratch area in which a fused finish left the planes of the own slot (phases of the pairs form)
        # The pairs form with its exchanges hidden (round 5; nb_launch_ring_fold_phase): a round's worth of the pairs inside the
        # rank's own slot of step k + 1 runs while step k's all-gather lands; the second exchange leaves as soon as the sums of the
        # ranks in front are final, and the rank's own sums are made beside it.  ring_overlap=None: overlap decides (True: where the
        # shape allows); True: required; False: fold, exchange, finish, all-gather in sequence.
        can = bool(self.partners) and hasattr(self.backend, "ring_phased") and self.backend.ring_phased(self.params, self.n, self.first, self.count)
        if ring_overlap and not can:
            raise ValueError("ring_overlap=True: this shape does not run its step in phases (nb_ring_phased() == 0)")
        self.ring_overlap = can and bool(overlap if ring_overlap is None else ring_overlap)
        if self.partners:
            sb = self.backend.ring_scratch_bytes(self.params, self.n, self.first, self.count)
        elif self.overlap:
            sb = self.backend.scratch_bytes_phased(self.params, self.n, self.count, self.first, self.first + self.count) if self.count else 0
        else:
            sb = self.backend.scratch_bytes(self.params, self.n, self.count) if self.count else 0
        self.scratch = torch.empty((sb,), dtype=torch.uint8, device=self.device) if sb else None
        self.cur = 0
        self.steps_done = 0
        # how the two exchanges are issued (verify_exchanges() may move either to its fallback)
        self.gather_in_place = True      # RCCL all-gather with this rank's slot of the receive buffer as the send buffer
        self.ring_grouped = True         # the second exchange as ONE group of sends and receives (else: one group per distance)
        self.exchange_report = None
        # "collective": torch.distributed (RCCL over xGMI: all-gather, grouped send / receive); "peers": pulls over xGMI ordered by
        # stream value waits (nb_peers_*, include/nenbody.h: no collective kernel, no second stream) for the position replicas and
        # `sums` -- everything else (velocities on demand, the boids controller's staging buffer) stays collective
        self.exchange = "collective"
        self._peers = None
        self._peers_sums = False
        if exchange not in ("collective", "peers"):
            raise ValueError('exchange must be "collective" or "peers"')
        if exchange == "peers" and world > 1:
            if not self.setup_peers():
                raise _lib.NbError(_lib.NB_ERR_STATE, "exchange=\"peers\": the ranks could not map each other's buffers (" + str(self.peers_error) + ")")
            self.exchange = "peers"
        self.velfull = None      # boids only: replicas of ALL velocities (ping-pong), built on first use
        self._pvstage = None     # boids, world > 1: [world][pos slot | vel slot], what the one all-gather per step moves
        self.velfull_valid = False

    # -- the exchanges as pulls over xGMI (nb_peers_*) ----------------------------------------------------------------------------
    def setup_peers(self) -> bool:
        """Collective: every rank registers its two position replicas (and `sums`, where the pairs form is planned), the blobs of
        IPC handles travel through torch.distributed, every rank maps the others'.  True when EVERY rank succeeded (one all-reduce);
        on False nothing changed and ``peers_error`` says why on the ranks that failed."""
        torch, dist = self.torch, self.dist
        self.peers_error = None
        if self._peers is not None:
            return True
        lib = self.backend.lib if hasattr(self.backend, "lib") else None
        handle, ok = ctypes.c_void_p(), lib is not None and self.device.type == "cuda" and self.world <= 16
        every = None
        if not ok:
            self.peers_error = "needs the HIP backend, device buffers and at most 16 ranks"
        blob = ctypes.create_string_buffer(int(lib.nb_peers_blob_bytes())) if lib is not None else None
        if ok:
            with torch.cuda.device(self.device):
                bufs = [self.pos[0], self.pos[1]] + ([self.sums] if self.sums is not None else [])
                rc = lib.nb_peers_create(self.rank, self.world, ctypes.byref(handle))
                if rc == _lib.NB_OK:
                    ptrs = (ctypes.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
                    sizes = (ctypes.c_size_t * len(bufs))(*[b.numel() * b.element_size() for b in bufs])
                    rc = lib.nb_peers_export(handle, ptrs, sizes, len(bufs), blob)
                if rc != _lib.NB_OK:
                    ok, self.peers_error = False, lib.nb_peers_last_error(handle if handle else None).decode()
        # (every rank takes part in the exchange of blobs whether it succeeded or not: a collective must not be skipped by some)
        every = [None] * self.world
        dist.all_gather_object(every, blob.raw if ok else b"", group=self.group)
        if ok and all(len(e) == len(blob.raw) for e in every):
            joined = b"".join(every)
            with torch.cuda.device(self.device):
                rc = lib.nb_peers_import(handle, ctypes.create_string_buffer(joined, len(joined)))
            if rc != _lib.NB_OK:
                ok, self.peers_error = False, lib.nb_peers_last_error(handle).decode()
        elif ok:
            ok, self.peers_error = False, "another rank could not export its buffers"
        every_imported = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(every_imported, op=dist.ReduceOp.MIN, group=self.group)
        if bool(every_imported.item()):
            # first contact with a deadline: a signal / wait / one-record pull round on a stream of its own -- if a peer's word never
            # becomes visible, that stream alone stays blocked and the steps keep their collectives
            with torch.cuda.device(self.device):
                rc = lib.nb_peers_probe(handle, 10000)
            if rc != _lib.NB_OK:
                ok, self.peers_error = False, lib.nb_peers_last_error(handle).decode()
        elif ok:
            ok, self.peers_error = False, "another rank could not map the buffers"
        on_device = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        if not bool(t.item()):
            if handle:
                lib.nb_peers_destroy(handle)
            self.peers_error = self.peers_error or "another rank failed"
            return False
        self._peers, self._peers_sums = handle, self.sums is not None
        return True

    def _peers_call(self, rc) -> None:
        if rc != _lib.NB_OK:
            raise _lib.NbError(rc, self.backend.lib.nb_peers_last_error(self._peers).decode())

    class _Pull:
        """an exchange by pulls whose signal is out: ``wait()`` issues the stream waits on the peers' flag words and the copy kernel"""

        def __init__(self, issue):
            self._issue = issue

        def wait(self):
            if self._issue is not None:
                self._issue()
                self._issue = None

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    # -- the exchange: every rank contributes its slot of `buf` and receives the others -----------------------
    def _all_gather_slots(self, buf, async_op: bool = False, slot: Optional[int] = None):
        slot = self.slot if slot is None else slot
        lo = self.rank * slot
        mine = buf[lo:lo + slot]
        if self.exchange == "peers" and slot == self.slot and (buf is self.pos[0] or buf is self.pos[1]):
            # pulls: the signal NOW (behind what the current stream holds: the finish kernel), the waits and the copy at wait()
            lib, idx = self.backend.lib, 0 if buf is self.pos[0] else 1
            with self.torch.cuda.device(self.device):
                self._peers_call(lib.nb_peers_signal(self._peers, 0, self._stream()))

            def issue():
                with self.torch.cuda.device(self.device):
                    self._peers_call(lib.nb_peers_gather(self._peers, 0, idx, self.slot * 16, self._stream()))

            pull = ShardedScene._Pull(issue)
            if async_op:
                return pull
            pull.wait()
            return None
        if self.dist.get_backend(self.group) == "nccl":
            # RCCL, in place: the send buffer is this rank's slot of the receive buffer (fallback: a copy of the slot)
            return self.dist.all_gather_into_tensor(buf, mine if self.gather_in_place else mine.clone(), group=self.group, async_op=async_op)
        elif buf.device.type == "cpu":
            self.dist.all_gather_into_tensor(buf, mine.clone(), group=self.group)
        else:
            # rehearsal path (gloo with device buffers, e.g. several ranks sharing one GPU): stage through the host
            full = self.torch.empty(buf.shape, dtype=buf.dtype)
            self.dist.all_gather_into_tensor(full, mine.cpu(), group=self.group)
            buf.copy_(full)
        return None  # these paths complete before returning

    # -- the pairs form's second exchange: chunk d of `sums` goes to rank + d, chunk d - 1 of `recv` comes from rank - d --------
    def _ring_exchange_start(self):
        """issues the exchange; returns what ``_ring_exchange_wait`` needs.  RCCL: the sends and receives run on the
        communicator's stream behind what the current stream holds NOW (the sums of the ranks in front), so launches that
        follow on the current stream overlap with them; the other paths complete before returning."""
        dist, S = self.dist, self.count
        if self.exchange == "peers" and self._peers_sums:
            lib = self.backend.lib
            with self.torch.cuda.device(self.device):
                self._peers_call(lib.nb_peers_signal(self._peers, 1, self._stream()))   # my chunks are final

            def issue():   # wait for the D ranks behind; one kernel copies their chunks for me
                with self.torch.cuda.device(self.device):
                    self._peers_call(lib.nb_peers_ring(self._peers, 1, 2, self.recv.data_ptr(), S * 16, self.partners, self._stream()))

            return [ShardedScene._Pull(issue)], None
        on_host = dist.get_backend(self.group) != "nccl" and self.sums.device.type != "cpu"
        sums = self.sums.cpu() if on_host else self.sums   # rehearsal path (gloo with device buffers): stage through the host
        recv = self.torch.empty(self.recv.shape, dtype=self.recv.dtype) if on_host else self.recv
        ops = []
        for d in range(1, self.partners + 1):
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            if self.group is not None:
                to, frm = dist.get_global_rank(self.group, to), dist.get_global_rank(self.group, frm)
            ops.append(dist.P2POp(dist.isend, sums[d * S:(d + 1) * S], to, self.group, tag=d))
            ops.append(dist.P2POp(dist.irecv, recv[(d - 1) * S:d * S], frm, self.group, tag=d))
        if self.ring_grouped:
            reqs = dist.batch_isend_irecv(ops)
        else:   # fallback: one group per distance d -- every rank sends to rank + d and receives from rank - d, then the next d
            reqs = []
            for i in range(0, len(ops), 2):
                reqs += dist.batch_isend_irecv(ops[i:i + 2])
        return reqs, (recv if on_host else None)

    def _ring_exchange_wait(self, started) -> None:
        reqs, staged = started
        for req in reqs:
            req.wait()   # RCCL: the current stream waits, not the host
        if staged is not None:
            self.recv.copy_(staged)

    def _ring_exchange(self) -> None:
        self._ring_exchange_wait(self._ring_exchange_start())

    def verify_exchanges(self) -> dict:
        """Before the first step of a world > 1: run both exchanges ONCE on a known per-rank pattern and check, on every rank, that
        every record arrived where the step will read it (collective; one all-reduce per check so that all ranks agree).  The
        in-place all-gather and the grouped sends / receives of the pairs form had run on a one-rank communicator and over gloo
        only when this was written (no multi-GPU box in the build); a mismatch moves the exchange to its fallback -- the all-gather
        from a copy of the slot, the second exchange as one group per distance, and, should that fail too, the ordered fold with its
        one exchange -- without restarting anything.  Returns (and keeps as ``exchange_report``) which paths the steps will take.
        Buffers used: the scratch side of the position ping-pong, ``sums`` / ``recv`` (every step overwrites them)."""
        torch, dist = self.torch, self.dist
        rep = {"world": self.world, "all_gather": None, "ring_exchange": None, "verified": False}
        if self.world == 1:
            self.exchange_report = rep
            return rep
        self._wait_pending()
        on_device = dist.get_backend(self.group) == "nccl"

        def agree(bad: bool) -> bool:   # True when ANY rank saw a mismatch
            t = torch.tensor([1 if bad else 0], dtype=torch.int32, device=self.device if on_device else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            return bool(t.item())

        def pattern(rank, records, salt):   # four words per record that name the sending rank, the record and the exchange
            i = torch.arange(records, dtype=torch.float32, device=self.device)
            return torch.stack([i + 1000.0 * rank + salt, -i, torch.full_like(i, float(rank)), torch.full_like(i, float(salt))], dim=1)

        # -- the all-gather: every slot of the buffer must hold its rank's pattern
        buf = self.pos[self.cur ^ 1]
        # (with pulls over xGMI the first attempt is the pulls; a mismatch sends BOTH exchanges back to the collectives.  The pulls are
        # checked TWICE from the same buffers with different patterns: a reader that kept lines of a peer's memory in a cache would pass
        # the first round and show the first pattern again in the second.  The agreement between the rounds is also the barrier that
        # lets a rank rewrite a buffer its peers have pulled from.)
        for attempt in (("peers",) if self.exchange == "peers" else ()) + ("in_place", "out_of_place"):
            if attempt != "peers":
                self.exchange = "collective"
            self.gather_in_place = attempt != "out_of_place"
            bad = False
            for salt in (7.0, 9.0) if attempt == "peers" else (7.0,):
                want = torch.cat([pattern(r, self.slot, salt) for r in range(self.world)])
                buf.fill_(float("nan"))
                buf[self.rank * self.slot:(self.rank + 1) * self.slot] = want[self.rank * self.slot:(self.rank + 1) * self.slot]
                self._all_gather_slots(buf)
                bad = agree(not torch.equal(buf, want))
                if bad:
                    break
            if not bad:
                rep["all_gather"] = attempt
                break
        buf.zero_()
        if rep["all_gather"] is None:
            raise _lib.NbError(_lib.NB_ERR_STATE, "verify_exchanges: the all-gather does not deliver every rank's slot, in place or from a copy")
        # -- the pairs form's second exchange: chunk d - 1 of recv must hold chunk d of rank - d's sums
        if self.partners:
            S = self.count
            for attempt in (("peers",) if self.exchange == "peers" and self._peers_sums else ()) + ("grouped", "per_distance"):
                if attempt != "peers" and self.exchange == "peers":
                    self.exchange = "collective"   # (the all-gather goes back too: one kind of exchange per step; it was verified above as pulls, the collective is re-verified by the next call)
                self.ring_grouped = attempt != "per_distance"
                bad = False
                for salt in (0.0, 16.0) if attempt == "peers" else (0.0,):   # (pulls: twice from the same buffers, as above)
                    want = torch.cat([pattern((self.rank - d) % self.world, S, salt + d) for d in range(1, self.partners + 1)])
                    self.sums[:S].zero_()
                    for d in range(1, self.partners + 1):
                        self.sums[d * S:(d + 1) * S] = pattern(self.rank, S, salt + d)
                    self.recv.fill_(float("nan"))
                    self._ring_exchange()
                    bad = agree(not torch.equal(self.recv, want))
                    if bad:
                        break
                if not bad:
                    rep["ring_exchange"] = attempt
                    break
            if rep["ring_exchange"] is None:   # neither form delivers: the ordered fold and its one exchange
                rep["ring_exchange"] = "disabled"
                self.partners, self.ring_overlap = 0, False
                self.sums = self.recv = None
                sb = self.backend.scratch_bytes(self.params, self.n, self.count) if self.count else 0
                self.scratch = torch.empty((sb,), dtype=torch.uint8, device=self.device) if sb else None
        rep["verified"] = True
        self.exchange_report = rep
        return rep

    def _wait_pending(self) -> None:
        """overlap: make the current stream wait for the exchange in flight (no host wait with RCCL)"""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    # -- boids: update_instance_boids (main.rs:443-526) reads every old velocity, so velocities are replicated and
    #    gathered like positions ----------------------------------------------------------------------------------
    def step_boids(self, params=None, split: bool = False) -> None:
        """One boids step.  split=True: the j range in slices (``nb_launch_boids_step_split``) -- the reference's neighbour sets
        and counts with reassociated sums, for shards whose bodies alone cannot fill the chip; default: bit-identical."""
        torch = self.torch
        self._wait_pending()
        self._own_ready = None
        bp = params if params is not None else _lib.default_boids_params()
        boids = self.backend.boids_step
        if split and self.count and hasattr(self.backend, "boids_step_split"):
            sb = self.backend.boids_split_scratch_bytes(bp, self.n, self.count)
            if getattr(self, "_boids_partial", None) is None or self._boids_partial.numel() < sb:
                self._boids_partial = torch.empty((sb,), dtype=torch.uint8, device=self.device)

            def boids(p_, n_, f_, c_, a_, b_, c2_, d_):
                self.backend.boids_step_split(p_, n_, f_, c_, a_, b_, c2_, d_, self._boids_partial)
        lo = self.rank * self.slot
        if self.velfull is None:
            padded = self.slot * self.world
            self.velfull = [torch.zeros((padded, 4), dtype=torch.float32, device=self.device) for _ in range(2)]
        if not self.velfull_valid:  # (re)build the replica from the local velocities (n-body steps keep only those)
            vf = self.velfull[self.cur]
            if self.count:
                vf[lo:lo + self.count] = self.vel[: self.count]
            if self.world > 1:
                self._all_gather_slots(vf)
            self.velfull_valid = True
        psrc, pdst = self.pos[self.cur], self.pos[self.cur ^ 1]
        vsrc, vdst = self.velfull[self.cur], self.velfull[self.cur ^ 1]
        if self.world > 1:
            # ONE exchange per step for both outputs: every rank's slot of the staging buffer is [slot position records | slot
            # velocity records]; the kernel writes this rank's new records straight into its slot (views offset so that record
            # first + l lands there: first == rank * slot), the slots are gathered in place, and the replicas are dealt out of it
            if self._pvstage is None:
                self._pvstage = torch.zeros((self.world * 2 * self.slot, 4), dtype=torch.float32, device=self.device)
            st = self._pvstage
            if self.count:
                boids(bp, self.n, self.first, self.count, psrc, vsrc, st[self.rank * self.slot:], st[self.rank * self.slot + self.slot:])
            self._all_gather_slots(st, slot=2 * self.slot)
            both = st.view(self.world, 2, self.slot, 4)
            pdst.view(self.world, self.slot, 4).copy_(both[:, 0])
            vdst.view(self.world, self.slot, 4).copy_(both[:, 1])
        elif self.count:
            boids(bp, self.n, self.first, self.count, psrc, vsrc, pdst, vdst)
        if self.count:
            self.vel[: self.count] = vdst[lo:lo + self.count]  # keep the local velocities current for n-body steps
        self.cur ^= 1
        self.steps_done += 1

    # -- one step: local update, then the exchange ------------------------------------------------------
    def step(self) -> None:
        src, dst = self.pos[self.cur], self.pos[self.cur ^ 1]
        # (did the fused finish of the step before leave the own slot's planes in THIS scratch area?  Every other form of a step, and
        # whatever rewrites the positions, uses it its own way)
        ready, self._own_ready = self._own_ready is not None and self._own_ready is self.scratch, None
        if self.partners and self.ring_overlap:
            # src's other slots may still be landing; this rank's own slot of src was written by its own last finish
            # The finish is the fused one: it leaves the planes of the new own slot in the scratch area, so every step but the first of a
            # run starts with its sweep.  Pulls run on this stream -- nothing could run beside them: the finish adds the rank's own records
            # itself; a collective runs on the communicator's stream, and NB_RING_SUMS keeps running beside it.
            be, a = self.backend, (self.params, self.n, self.first, self.count)
            be.ring_fold_phase(*a, _lib.NB_RING_OWN_READY if ready else _lib.NB_RING_OWN, src, self.sums, self.scratch)
            self._wait_pending()
            be.ring_fold_phase(*a, _lib.NB_RING_REST, src, self.sums, self.scratch)     # the sums of the ranks in front are final
            started = self._ring_exchange_start()
            in_stream = self.exchange == "peers" and self._peers_sums
            if not in_stream:
                be.ring_fold_phase(*a, _lib.NB_RING_SUMS, src, self.sums, self.scratch)     # ... the rank's own beside the exchange
            self._ring_exchange_wait(started)
            be.ring_finish_phase(*a, src, dst, self.vel, None if in_stream else self.sums, self.recv, self.scratch)
            self._own_ready = self.scratch
            self._pending = self._all_gather_slots(dst, async_op=True)
        elif self.partners:
            self._wait_pending()
            self.backend.ring_fold(self.params, self.n, self.first, self.count, src, self.sums, self.scratch)
            self._ring_exchange()
            self.backend.ring_finish(self.params, self.n, self.first, self.count, src, dst, self.vel, self.sums, self.recv)
            self._all_gather_slots(dst)
        elif self.overlap:
            # src's other slots may still be landing; this rank's own slot of src was written by its own last step
            lo, hi = self.first, self.first + self.count
            if self.count:
                self.backend.step_phase(self.params, self.n, self.first, self.count, lo, hi, _lib.NB_PHASE_RANGE, src, dst, self.vel,
                                        self.scratch)
            self._wait_pending()
            if self.count:
                self.backend.step_phase(self.params, self.n, self.first, self.count, lo, hi, _lib.NB_PHASE_REST, src, dst, self.vel,
                                        self.scratch)
            self._pending = self._all_gather_slots(dst, async_op=True)
        else:
            self._wait_pending()
            if self.count:
                self.backend.step(self.params, self.n, self.first, self.count, src, dst, self.vel, self.scratch)
            if self.world > 1:
                self._all_gather_slots(dst)
        self.velfull_valid = False  # the velocity replica (boids only) no longer matches the local velocities
        self.cur ^= 1
        self.steps_done += 1

    def step_n(self, k: int) -> None:
        for _ in range(int(k)):
            self.step()

    def choose_form(self, steps: int = 4, warm: int = 2) -> str:
        """FAST where the pairs form is planned: time ``steps`` steps of every form this shape can take -- the pairs form with its
        two exchanges in sequence ("pairs"), the same with the exchanges behind compute ("pairs_overlapped", where
        ``nb_ring_phased``), the ordered fold with its one exchange ("ordered") -- on the current state, every rank taking the
        SLOWEST rank's time (one all-reduce), keep the fastest for the steps to come, and put the state back.  What an exchange
        costs between real GPUs is not known ahead of time (the library's own line is drawn from one-GPU
        timings and a one-rank communicator); this asks the machine.  Collective: every rank calls it, every rank gets the same
        answer (``form_times``: seconds per step of each candidate).  Other shapes return "ordered" at once."""
        torch = self.torch
        self.form_times = None
        if not self.partners:
            return "ordered"
        self._wait_pending()
        saved = (self.pos[0].clone(), self.pos[1].clone(), self.vel.clone(), self.cur, self.steps_done, self.velfull_valid)
        sb = self.backend.scratch_bytes(self.params, self.n, self.count) if self.count else 0
        phased = hasattr(self.backend, "ring_phased") and self.backend.ring_phased(self.params, self.n, self.first, self.count)
        # candidate -> (partners, scratch, ring_overlap)
        cands = {"pairs": (self.partners, self.scratch, False)}
        if phased:
            cands["pairs_overlapped"] = (self.partners, self.scratch, True)
        cands["ordered"] = (0, torch.empty((sb,), dtype=torch.uint8, device=self.device) if sb else None, False)

        def fence():
            self._wait_pending()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.dist.barrier(group=self.group)

        def restore():
            self._wait_pending()
            self.pos[0].copy_(saved[0])
            self.pos[1].copy_(saved[1])
            self.vel.copy_(saved[2])
            self.cur, self.steps_done, self.velfull_valid = saved[3], saved[4], saved[5]
            self._own_ready = None

        names, times = list(cands), []
        for name in names:
            self.partners, self.scratch, self.ring_overlap = cands[name]
            self.step_n(warm)
            fence()
            t0 = time.perf_counter()
            self.step_n(steps)
            fence()
            times.append((time.perf_counter() - t0) / max(1, steps))
            restore()
        on_device = self.dist.get_backend(self.group) == "nccl"
        t = torch.tensor(times, dtype=torch.float64, device=self.device if on_device else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        self.form_times = dict(zip(names, (float(x) for x in t.tolist())))
        best = min(names, key=lambda k: (self.form_times[k], names.index(k)))   # (a tie keeps the earlier, simpler form)
        self.partners, self.scratch, self.ring_overlap = cands[best]
        if best == "ordered":
            self.sums = self.recv = None
        return best

    def choose_exchange(self, steps: int = 6, warm: int = 2) -> str:
        """Which exchange is faster on THIS machine for the form the steps take now: the collectives (RCCL's all-gather and grouped
        send / receive) or pulls over xGMI ordered by stream value waits (``setup_peers``)?  Verifies the pulls on a pattern first,
        times ``steps`` steps each way on the state in hand (slowest rank; collective), keeps the faster, puts the state back.
        Returns "peers" or "collective" (``exchange_times``: seconds per step of each; None where the ranks could not map each
        other's buffers -- another node, no IPC)."""
        torch = self.torch
        self.exchange_times = None
        if self.world == 1:
            return self.exchange
        self._wait_pending()
        if not self.setup_peers():
            return self.exchange
        before = self.exchange
        self.exchange = "peers"
        if self.verify_exchanges()["all_gather"] != "peers" or self.exchange != "peers":   # the pattern did not arrive through the pulls
            self.exchange_times = {"peers": None, "collective": None}
            return self.exchange
        saved = (self.pos[0].clone(), self.pos[1].clone(), self.vel.clone(), self.cur, self.steps_done, self.velfull_valid)
        times = {}
        for kind in ("collective", "peers"):
            self.exchange = kind
            self.step_n(warm)
            self._wait_pending()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.dist.barrier(group=self.group)
            t0 = time.perf_counter()
            self.step_n(steps)
            self._wait_pending()
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.dist.barrier(group=self.group)
            times[kind] = (time.perf_counter() - t0) / max(1, steps)
            self._wait_pending()
            self.pos[0].copy_(saved[0])
            self.pos[1].copy_(saved[1])
            self.vel.copy_(saved[2])
            self.cur, self.steps_done, self.velfull_valid = saved[3], saved[4], saved[5]
            self._own_ready = None
        on_device = self.dist.get_backend(self.group) == "nccl"
        t = torch.tensor([times["collective"], times["peers"]], dtype=torch.float64, device=self.device if on_device else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        self.exchange_times = {"collective": float(t[0]), "peers": float(t[1])}
        self.exchange = "peers" if self.exchange_times["peers"] < self.exchange_times["collective"] else "collective"
        del before
        return self.exchange

    def close(self) -> None:
        """drop the mappings of the other ranks' buffers (``setup_peers``); the scene keeps working over the collectives"""
        if getattr(self, "_peers", None):
            try:
                self._wait_pending()
                if self.device.type == "cuda":
                    self.torch.cuda.synchronize(self.device)
                self.backend.lib.nb_peers_destroy(self._peers)
            finally:
                self._peers = None
                self.exchange = "collective"

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        """Wait for the queued steps; raises NbError (NB_ERR_STATE) if a kernel reported a failure (``nb_launch_status``)."""
        self._wait_pending()
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)
            if self.backend.name == "hip":
                with self.torch.cuda.device(self.device):
                    check(self.backend.lib.nb_launch_status(self.torch.cuda.current_stream(self.device).cuda_stream))

    # -- state access -------------------------------------------------------------------------------------
    def positions(self) -> np.ndarray:
        """All n positions (every rank holds the replica), shape (n, 3)."""
        self._wait_pending()
        return self.pos[self.cur][: self.n, :3].cpu().numpy().copy()

    def local_velocities(self) -> np.ndarray:
        return self.vel[: self.count, :3].cpu().numpy().copy()

    def velocities(self) -> np.ndarray:
        """All n velocities, gathered on demand (not part of the per-step exchange)."""
        if self.world == 1:
            return self.local_velocities()
        on_host = self.dist.get_backend(self.group) != "nccl"
        dev = "cpu" if on_host else self.device
        slot_v = self.torch.zeros((self.slot, 4), dtype=self.torch.float32, device=dev)
        slot_v[: self.count] = self.vel[: self.count].to(dev)
        full = self.torch.zeros((self.slot * self.world, 4), dtype=self.torch.float32, device=dev)
        self.dist.all_gather_into_tensor(full, slot_v, group=self.group)
        return full[: self.n, :3].cpu().numpy().copy()

    def local_instances(self) -> np.ndarray:
        """Model matrices (main.rs:437-439) of this rank's bodies, shape (count, 4, 4)."""
        self._wait_pending()
        inst = self.torch.zeros((max(self.count, 1), 16), dtype=self.torch.float32, device=self.device)
        if self.count:
            mine = self.pos[self.cur][self.first:self.first + self.count]
            self.backend.instances(self.count, mine, self.vel, inst)
        return inst[: self.count].cpu().numpy().reshape(self.count, 4, 4).copy()


def comm_id() -> bytes:
    """A fresh RCCL unique id (``nb_comm_id``): make it on one rank, hand the bytes to the others by any channel."""
    buf = ctypes.create_string_buffer(_lib.NB_COMM_ID_BYTES)
    check(_lib.load().nb_comm_id(buf))
    return buf.raw


class NativeShard:
    """One rank's share of a scene with the host side inside libnenbody_hip.so (``nb_shard_*``, include/nenbody.h).

    The same decomposition as :class:`ShardedScene` -- index ranges, one exchange of positions per step (velocities too
    for the boids controller) -- without torch: the exchange is RCCL (``comm_id`` = the bytes of :func:`comm_id`, made
    on one rank), pulls over xGMI (``peers``: a collective callable blob -> all ranks' blobs; round 5) or ``gather``, a callable ``(buf_ptr, slot_bytes, rank, world, stream_ptr) -> None`` that completes the
    all-gather of the device buffer at ``buf_ptr``.  FAST with equal ranks of whole blocks takes the pairs form (every unordered
    pair once, ``nb_launch_ring_fold``) where its second exchange exists: RCCL's send / receive, or ``ring``, a callable
    ``(send_ptr, recv_ptr, chunk_bytes, partners, rank, world, stream_ptr) -> None`` beside ``gather`` (chunk d - 1 of ``send``
    goes to rank + d, chunk d - 1 of ``recv`` comes from rank - d); ``pairs=False`` keeps the ordered fold.  ``partners`` tells
    which a step will take (0: the ordered fold).  This is what a Rust or C++ host binds; it is wrapped here so the tests can
    drive it.
    """

    def __init__(self, positions, velocities, params: Optional[NbParams] = None, *, rank: int = 0, world: int = 1,
                 comm_id: Optional[bytes] = None, gather=None, overlap: bool = False, ring=None, pairs: Optional[bool] = None,
                 boids_split: bool = False, peers=None):
        lib = _lib.load()
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        vel = np.ascontiguousarray(velocities, dtype=np.float32)
        if pos.ndim != 2 or pos.shape[1] != 3 or pos.shape != vel.shape:
            raise ValueError("positions and velocities must both have shape (n, 3)")
        self.n, self.rank, self.world = len(pos), int(rank), int(world)
        self.params = params if params is not None else _lib.default_params()
        self._lib = lib
        self._sh = ctypes.c_void_p()
        check(lib.nb_shard_create(self.n, self.rank, self.world, ctypes.byref(self.params), ctypes.byref(self._sh)))
        try:
            first, count = ctypes.c_uint32(), ctypes.c_uint32()
            self._check(lib.nb_shard_range(self._sh, ctypes.byref(first), ctypes.byref(count)))
            self.first, self.count = first.value, count.value
            self._gather_keepalive = None
            if comm_id is not None:
                if len(comm_id) != _lib.NB_COMM_ID_BYTES:
                    raise ValueError(f"comm_id must be {_lib.NB_COMM_ID_BYTES} bytes")
                self._check(lib.nb_shard_use_rccl(self._sh, ctypes.create_string_buffer(comm_id, len(comm_id))))
            elif gather is not None:
                def trampoline(_user, buf, slot_bytes, rank_, world_, stream):
                    try:
                        gather(buf, slot_bytes, rank_, world_, stream)
                        return 0
                    except Exception:  # pragma: no cover - reported as NB_ERR_STATE by the library
                        import traceback

                        traceback.print_exc()
                        return 1

                self._gather_keepalive = _lib.GATHER_FN(trampoline)
                self._check(lib.nb_shard_use_gather(self._sh, self._gather_keepalive, None))
            self._ring_keepalive = None
            if ring is not None:
                def ring_trampoline(_user, send, recv, chunk_bytes, partners, rank_, world_, stream):
                    try:
                        ring(send, recv, chunk_bytes, partners, rank_, world_, stream)
                        return 0
                    except Exception:  # pragma: no cover - reported as NB_ERR_STATE by the library
                        import traceback

                        traceback.print_exc()
                        return 1

                self._ring_keepalive = _lib.RING_FN(ring_trampoline)
                self._check(lib.nb_shard_use_ring(self._sh, self._ring_keepalive, None))
            if peers is not None:
                # both exchanges as pulls over xGMI (nb_shard_peer_export / _import): `peers` is a collective callable that takes this
                # rank's blob (bytes) and returns every rank's, rank-major, concatenated -- any channel the host has
                blob = ctypes.create_string_buffer(int(lib.nb_peers_blob_bytes()))
                self._check(lib.nb_shard_peer_export(self._sh, blob))
                every = peers(blob.raw)
                if len(every) != self.world * len(blob.raw):
                    raise ValueError("peers(blob) must return world blobs, rank-major, concatenated")
                self._check(lib.nb_shard_peer_import(self._sh, ctypes.create_string_buffer(every, len(every))))
            if pairs is not None:
                self._check(lib.nb_shard_set_pairs(self._sh, 1 if pairs else 0))
            if boids_split:  # the boids step's j range in slices: the reference's neighbour sets and counts, reassociated sums
                self._check(lib.nb_shard_set_boids_split(self._sh, 1))
            if overlap:  # FAST only; a STRICT shard ignores it (nb_shard_set_overlap)
                self._check(lib.nb_shard_set_overlap(self._sh, 1))
            self._check(lib.nb_shard_upload(self._sh, pos.ctypes.data, vel.ctypes.data))
        except Exception:
            self.close()
            raise

    def _check(self, rc: int) -> None:
        if rc != _lib.NB_OK:
            raise _lib.NbError(rc, self._lib.nb_shard_last_error(self._sh).decode())

    @property
    def partners(self) -> int:
        """D of the pairs form a step will take (``nb_shard_pairs_partners``); 0: the ordered fold and its one exchange."""
        return int(self._lib.nb_shard_pairs_partners(self._sh))

    @property
    def pairs_overlapped(self) -> bool:
        """does a step take the pairs form in phases, both exchanges on a second stream (``nb_shard_pairs_overlapped``)?"""
        return int(self._lib.nb_shard_pairs_overlapped(self._sh)) == 1

    def use_peers(self, on: bool) -> None:
        """switch between the pulls over xGMI (after ``peers=`` at construction) and the exchange chosen before"""
        self._check(self._lib.nb_shard_use_peers(self._sh, 1 if on else 0))

    def verify_exchanges(self):
        """both exchanges once on a known pattern, checked on every rank (``nb_shard_verify_exchanges``; collective): returns
        (gather_path, ring_path) -- 0 / 1 / 2: in place / from a copy / pulled over xGMI; -1 / 0 / 1 / 2 / 3: no pairs form / one group /
        one group per distance / dropped for the ordered fold / pulled over xGMI"""
        g, r = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.nb_shard_verify_exchanges(self._sh, ctypes.byref(g), ctypes.byref(r)))
        return g.value, r.value

    def choose_form(self, steps: int = 4):
        """times every form a FAST step of this shard can take on this machine and keeps the fastest (``nb_shard_choose_form``;
        collective): returns (chosen, [ms per step of the ordered fold, the pairs form, the pairs form overlapped]; < 0: not offered)"""
        c, ms = ctypes.c_int(), (ctypes.c_double * 3)()
        self._check(self._lib.nb_shard_choose_form(self._sh, int(steps), ctypes.byref(c), ms))
        return c.value, list(ms)

    def upload(self, positions, velocities) -> None:
        """Replaces the state (all n positions, all n velocities; the rank keeps its own range of the latter)."""
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        vel = np.ascontiguousarray(velocities, dtype=np.float32)
        if pos.shape != (self.n, 3) or vel.shape != (self.n, 3):
            raise ValueError("positions and velocities must both have shape (n, 3)")
        self._check(self._lib.nb_shard_upload(self._sh, pos.ctypes.data, vel.ctypes.data))

    def step(self, k: int = 1) -> None:
        self._check(self._lib.nb_shard_step(self._sh, int(k)))

    def step_boids(self, k: int = 1, params=None) -> None:
        self._check(self._lib.nb_shard_step_boids(self._sh, int(k), ctypes.byref(params) if params is not None else None))

    def sync(self) -> None:
        self._check(self._lib.nb_shard_sync(self._sh))

    def positions(self) -> np.ndarray:
        """All n positions (every rank holds the replica)."""
        out = np.zeros((self.n, 3), np.float32)
        self._check(self._lib.nb_shard_download(self._sh, out.ctypes.data, None, None))
        return out

    def local_velocities(self) -> np.ndarray:
        out = np.zeros((self.count, 3), np.float32)
        self._check(self._lib.nb_shard_download(self._sh, None, out.ctypes.data if self.count else None, None))
        return out

    def local_instances(self) -> np.ndarray:
        out = np.zeros((self.count, 4, 4), np.float32)
        self._check(self._lib.nb_shard_download(self._sh, None, None, out.ctypes.data if self.count else None))
        return out

    def close(self) -> None:
        if getattr(self, "_sh", None) is not None and self._sh:
            self._lib.nb_shard_destroy(self._sh)
            self._sh = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass
