"""A compute backend for nenbody_amd.dist.ShardedScene that runs the CPU oracle on CPU tensors.

TEST-ONLY: it lets the world_size > 1 orchestration (partition, per-step all-gather, buffer ping-pong) be
exercised under the gloo backend on a machine with no GPU.  The product backend is HipBackend.
"""
import numpy as np
import torch

import oracle


class OracleBackend:
    name = "oracle-test"

    def scratch_bytes(self, params, n_total, count):
        return 0

    def step(self, params, n_total, first, count, pos_in, pos_out, vel, scratch):
        old = pos_in[:n_total, :3].contiguous().numpy()
        v = vel[:count, :3].contiguous().numpy()
        p_new, v_new = oracle.step_range(old, v, first, count, np.float32(params.dt), np.float32(params.G),
                                         np.float32(params.bias))
        pos_out[first:first + count, :3] = torch.from_numpy(p_new)
        pos_out[first:first + count, 3] = 0
        vel[:count, :3] = torch.from_numpy(v_new)

    # -- a FAST step in two phases (nb_launch_step_phase): the sums of records [j_lo, j_hi) first, the rest later --------
    def scratch_bytes_phased(self, params, n_total, count, j_lo, j_hi):
        return 16 * count

    @staticmethod
    def _partial(params, old, first, count, js):
        """sum over records js of ((p_j - p_n) * G) / (|p_j - p_n|^2 + bias) for bodies [first, first+count): numpy binary32,
        any order -- this backend tests the ORCHESTRATION (what is read when), FAST tolerances apply"""
        d = old[js][None, :, :] - old[first:first + count][:, None, :]
        r2 = (d * d).sum(axis=2, dtype=np.float32) + np.float32(params.bias)
        return ((d * np.float32(params.G)) / r2[:, :, None]).sum(axis=1, dtype=np.float32)

    def step_phase(self, params, n_total, first, count, j_lo, j_hi, phase, pos_in, pos_out, vel, scratch):
        old = pos_in[:n_total, :3].contiguous().numpy()
        part = scratch.view(torch.float32).reshape(count, 4)
        if phase == 0:
            part[:, :3] = torch.from_numpy(self._partial(params, old, first, count, np.arange(j_lo, j_hi)))
            return
        rest = np.concatenate([np.arange(0, j_lo), np.arange(j_hi, n_total)])
        a = part[:, :3].numpy() + self._partial(params, old, first, count, rest)
        v = vel[:count, :3].numpy() + a * np.float32(params.dt)
        pos_out[first:first + count, :3] = torch.from_numpy(v + old[first:first + count])
        pos_out[first:first + count, 3] = 0
        vel[:count, :3] = torch.from_numpy(v)

    # -- FAST on shards, every unordered pair once (nb_launch_ring_fold / nb_launch_ring_finish): the same decomposition at a block
    #    size of ONE body -- body I evaluates its pairs with the h(I) bodies that follow it on the ring, keeps its own halves and
    #    files the others under the offset of the body they belong to; numpy binary32, FAST tolerances apply -----------------------
    @staticmethod
    def _fwd(i, n):
        return (n - 1) // 2 if n % 2 else n // 2 - 1 + (1 if i < n // 2 else 0)

    def ring_partners(self, params, n_total, first, count):
        if params.mode != 1 or count == 0 or count >= n_total or n_total % count or first % count:
            return 0
        hmax = (n_total - 1) // 2 if n_total % 2 else n_total // 2
        return -(-(count + hmax) // count) - 1

    def ring_scratch_bytes(self, params, n_total, first, count):
        return 3 * count * 16    # the phases keep two rows of count records here between their calls, the fused finish a third

    # the step in PHASES (nb_launch_ring_fold_phase): 1 = pairs inside the rank's own slot (here: those of the first 70 % of its
    # bodies -- a part, as the library's one round of workgroups is), 2 = every other pair + the sums of the ranks in front, 3 = the
    # rank's own sums.  Phase 1 sees a snapshot whose OTHER slots are NaN: a read of a record that may still be in flight poisons
    # the result.
    def ring_phased(self, params, n_total, first, count):
        return self.ring_partners(params, n_total, first, count) > 0

    #    Phase 4 (NB_RING_OWN_READY) is phase 1 on the planes the fused finish of the step before left: here a third row of the
    #    scratch area holds the own slot's new positions, and phase 4 reads THEM -- never pos_in.
    def ring_fold_phase(self, params, n_total, first, count, phase, pos_in, sums, scratch):
        keep = scratch.view(torch.float32)[:3 * count * 4].reshape(3 * count, 4)
        own_acc, rest_acc, planes = keep[:count, :3].numpy(), keep[count:2 * count, :3].numpy(), keep[2 * count:, :3].numpy()
        if phase == 4:
            snap = torch.full((n_total, 4), float("nan"))
            snap[first:first + count, :3] = torch.from_numpy(planes.copy())
            pos_in, phase = snap, 1
        if phase == 3:
            sums[:count, :3] = torch.from_numpy(own_acc + rest_acc)
            sums[:count, 3] = 0
            return
        old = pos_in[:n_total, :3].contiguous().numpy().copy()
        head = (7 * count + 9) // 10     # bodies whose own-slot pairs phase 1 takes
        if phase == 1:
            mask = np.ones(n_total, bool)
            mask[first:first + count] = False
            old[mask] = np.nan
            own_acc[:] = 0
        else:
            rest_acc[:] = 0
        out = np.zeros((sums.shape[0], 3), np.float32)
        for l in range(count):
            i = first + l
            k = np.arange(self._fwd(i, n_total))
            in_head = (l + 1 + k < count) & (l < head)      # the other body is one of this rank's own, and the pair is phase 1's
            k = k[in_head] if phase == 1 else k[~in_head]
            if not len(k):
                continue
            js = (i + 1 + k) % n_total
            d = old[js] - old[i]
            t = d / ((d * d).sum(axis=1, dtype=np.float32) + np.float32(params.bias))[:, None]
            if phase == 1:
                own_acc[l] += t.sum(axis=0, dtype=np.float32)
                np.subtract.at(own_acc, l + 1 + k, t)
            else:
                rest_acc[l] += t.sum(axis=0, dtype=np.float32)
                np.subtract.at(out, l + 1 + k, t)
        if phase == 2:
            rest_acc[:] += out[:count]            # (own-slot pairs phase 1 left: their other halves stay in the rank)
            sums[count:, :3] = torch.from_numpy(out[count:])
            sums[count:, 3] = 0

    def ring_fold(self, params, n_total, first, count, pos_in, sums, scratch):
        old = pos_in[:n_total, :3].contiguous().numpy()
        out = np.zeros((sums.shape[0], 3), np.float32)
        for l in range(count):
            i = first + l
            js = (i + 1 + np.arange(self._fwd(i, n_total))) % n_total
            if not len(js):
                continue
            d = old[js] - old[i]
            t = d / ((d * d).sum(axis=1, dtype=np.float32) + np.float32(params.bias))[:, None]
            out[l] += t.sum(axis=0, dtype=np.float32)
            np.subtract.at(out, l + 1 + np.arange(len(js)), t)   # offsets behind this rank's first body: the ring, unrolled
        sums[:, :3] = torch.from_numpy(out)
        sums[:, 3] = 0

    def ring_finish(self, params, n_total, first, count, pos_in, pos_out, vel, sums, recv):
        old = pos_in[:n_total, :3].contiguous().numpy()
        a = sums[:count, :3].numpy().copy()
        for d in range(recv.shape[0] // count):
            a = a + recv[d * count:(d + 1) * count, :3].numpy()
        v = vel[:count, :3].numpy() + (a * np.float32(params.G)) * np.float32(params.dt)
        pos_out[first:first + count, :3] = torch.from_numpy(v + old[first:first + count])
        pos_out[first:first + count, 3] = 0
        vel[:count, :3] = torch.from_numpy(v)

    # the fused finish: sums None -> phase 3's addition happens here; the new own slot is left in the scratch area for phase 4
    def ring_finish_phase(self, params, n_total, first, count, pos_in, pos_out, vel, sums, recv, scratch):
        keep = scratch.view(torch.float32)[:3 * count * 4].reshape(3 * count, 4)
        if sums is None:
            sums = torch.zeros((count, 4))
            sums[:, :3] = torch.from_numpy(keep[:count, :3].numpy() + keep[count:2 * count, :3].numpy())
        self.ring_finish(params, n_total, first, count, pos_in, pos_out, vel, sums, recv)
        keep[2 * count:, :3] = pos_out[first:first + count, :3]

    def instances(self, count, pos, vel, inst):
        m = oracle.instances(pos[:count, :3].contiguous().numpy(), vel[:count, :3].contiguous().numpy())
        inst[:count] = torch.from_numpy(m.reshape(count, 16))

    def boids_step(self, params, n_total, first, count, pos_in, vel_in, pos_out, vel_out):
        bp = oracle.boids_params()
        for k in ("dt", "rule_1_distance", "rule_2_distance", "rule_3_distance", "rule_1_scale", "rule_2_scale", "rule_3_scale"):
            setattr(bp, k, getattr(params, k))
        op = pos_in[:n_total, :3].contiguous().numpy()
        ov = vel_in[:n_total, :3].contiguous().numpy()
        p_new, v_new = oracle.boids_step_range(op, ov, first, count, bp)
        pos_out[first:first + count, :3] = torch.from_numpy(p_new)
        pos_out[first:first + count, 3] = 0
        vel_out[first:first + count, :3] = torch.from_numpy(v_new)
        vel_out[first:first + count, 3] = 0
