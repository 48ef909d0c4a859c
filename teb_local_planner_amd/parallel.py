"""Multi-GPU sharding of the candidate batch and the path's only exchange step.

The candidate TEBs of HomotopyClassPlanner are independent units (reference
src/homotopy_class_planner.cpp:468 "independend of each other"): rank r owns a contiguous block of the B
candidates, the obstacle table / config are replicated, state strips never move between GPUs, and there is NO
data-path collective. Per plan() there is exactly one exchange: selectBestTeb
(src/homotopy_class_planner.cpp:564-667) = argmin of the (already hysteresis-scaled) costs with the lowest
index winning ties (strict '<' at :610). Each rank contributes its local winner as a 16-byte record
(cost f64, global index as f64) to an all-gather (RCCL over xGMI when the backend is "nccl", gloo in the CPU
tests); every rank then takes the lexicographic minimum, so all ranks agree without a second collective.
"""
import ctypes as C

import numpy as np


def shard_range(total, rank, world):
    """Contiguous block partition: the first (total % world) ranks get one extra candidate."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_best(costs, offset=0, last_best=-1, initial_plan=-1, hysteresis=1.0, prefer_initial=1.0):
    """selectBestTeb restricted to this rank's candidates; indices are GLOBAL (offset + local).
    Returns (cost, global_index) with cost = +inf and index = -1 when the rank owns nothing."""
    best_c, best_i = np.finfo(np.float64).max, -1
    for k, c in enumerate(np.asarray(costs, dtype=np.float64)):
        g = offset + k
        if g == last_best:
            c = c * hysteresis
        elif g == initial_plan:
            c = c * prefer_initial
        if c < best_c:
            best_c, best_i = c, g
    return best_c, best_i


# what a rank contributes when its local selection failed (bad handle, device mismatch, failed launch): it still enters the collective -
# its peers must not wait for ever - with a record no rank can pick (teb_amd_select_best_distributed does the same inside libteb_amd.so)
UNUSABLE_RECORD = (float(np.finfo(np.float64).max), -1)


def pick_global(records):
    """records: iterable of (cost, global_index). Lowest cost wins, ties -> lowest index; -1 entries ignored."""
    best_c, best_i = np.finfo(np.float64).max, -1
    for c, i in records:
        i = int(i)
        if i < 0:
            continue
        if c < best_c or (c == best_c and (best_i < 0 or i < best_i)):
            best_c, best_i = float(c), i
    return best_c, best_i


def select_best_distributed(cost, global_index, group=None, device=None):
    """All-gather of one (cost, index) record per rank; returns the same (cost, index) on every rank.

    Works with any initialised torch.distributed backend: "nccl" (= RCCL on ROCm; pass device="cuda") or "gloo"."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rec = torch.tensor([float(cost), float(global_index)], dtype=torch.float64, device=device)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec, group=group)
    allv = torch.stack(out).cpu().numpy()
    return pick_global((allv[k, 0], allv[k, 1]) for k in range(world))


COMM_ID_BYTES = 128


class RcclComm:
    """teb_amd_comm_t: the RCCL communicator the C-ABI's exchange runs on (include/teb_amd.h, multi-GPU section).

    rank 0 creates the 128-byte id (RcclComm.unique_id()) and ships it to the other ranks by any means; from_torch() uses an
    already initialised torch.distributed group for exactly that and nothing else - the exchange itself happens inside libteb_amd.so."""

    def __init__(self, unique_id, rank, world, device):
        from . import planner
        self._c = C.c_void_p(None)
        self.rank, self.world, self.device = rank, world, device
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        planner._chk(planner.lib().teb_amd_comm_create(buf, rank, world, device, C.byref(self._c)), "teb_amd_comm_create")

    @staticmethod
    def unique_id():
        from . import planner
        buf = C.create_string_buffer(COMM_ID_BYTES)
        planner._chk(planner.lib().teb_amd_comm_unique_id(buf), "teb_amd_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch(cls, device, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], rank, world, device)

    def close(self):
        if self._c:
            from . import planner
            planner.lib().teb_amd_comm_destroy(self._c)
            self._c = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def broadcast_band(owner_rank, band, capacity, local_status=0, group=None, device=None):
    """Mirror of teb_amd_broadcast_band (csrc/teb_amd.hip) on a torch.distributed group: round 1 an all-gather of (status, capacity) per
    rank - every rank learns of a peer's error or of differing capacities BEFORE the broadcast and all return the same verdict -, round
    2 the winner's strip [n | x | y | theta | dt | statistics] from its owner. band: (x, y, theta, dt, available, back_chi2) on the owner
    (ignored elsewhere). Returns (ok, band or None)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_rank(group) == owner_rank and local_status == 0:
        # the owner's own precondition goes into round 1 like any other error: a band that is missing or longer than the message would
        # otherwise raise on the owner alone, in front of the broadcast every peer has already entered
        if band is None or len(band[0]) > capacity or len(band[0]) < 1:
            local_status = 1
    rec = torch.tensor([float(local_status), float(capacity)], dtype=torch.float64, device=device)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec, group=group)
    allv = torch.stack(out).cpu().numpy()
    if (allv[:, 0] != 0).any() or (allv[:, 1] != capacity).any():
        return False, None
    msg = torch.zeros(3 + 4 * capacity, dtype=torch.float64, device=device)
    if dist.get_rank(group) == owner_rank:
        x, y, th, dt, avail, back = band
        n = len(x)
        msg[0] = n
        for k, a in enumerate((x, y, th)):
            msg[1 + k * capacity:1 + k * capacity + n] = torch.as_tensor(np.asarray(a, np.float64))
        msg[1 + 3 * capacity:1 + 3 * capacity + n - 1] = torch.as_tensor(np.asarray(dt, np.float64))
        msg[1 + 4 * capacity] = float(avail); msg[2 + 4 * capacity] = float(back)
    dist.broadcast(msg, src=owner_rank, group=group)
    m = msg.cpu().numpy()
    n = int(m[0])
    if n > capacity:
        return False, None
    return True, (m[1:1 + n].copy(), m[1 + capacity:1 + capacity + n].copy(), m[1 + 2 * capacity:1 + 2 * capacity + n].copy(),
                  m[1 + 3 * capacity:1 + 3 * capacity + max(n - 1, 0)].copy(), bool(m[1 + 4 * capacity]), float(m[2 + 4 * capacity]))


def sharded_plan_exchange(local_ok, local_record, local_bands, owner_of, capacity, fail_inside_selection=False, fail_before_broadcast=False,
                          group=None, device=None):
    """The collective sequence of HomotopyClassPlannerAmd::plan() in its sharded mode (host/teb_amd_hcp_backend.cpp), as the ranks of a
    torch.distributed group run it - what keeps a tick deadlock-free when ONE rank fails somewhere in its own work:
      1. every rank enters the selection all-gather; a rank whose exploration / upload / optimisation failed (local_ok False) sends the
         unusable record; a rank on which the selection call itself fails AFTER its record went out (fail_inside_selection) learns the
         peers' choice all the same;
      2. no rank holds a candidate (index < 0): every rank sees that and none enters the broadcast;
      3. otherwise every rank - the failed ones too - enters the broadcast of the winner's band, statistics included; a rank that fails
         BETWEEN selection and broadcast (fail_before_broadcast: e.g. the owner cannot read its winner back) says so in the broadcast's
         first round, and every rank returns not-ok together instead of waiting for a strip that never comes.
    local_record: (cost, global index) of this rank's best candidate; local_bands: {global index: band tuple}; owner_of(global index) -> rank.
    Returns (ok, global index, band): ok False on the rank that failed, the winner and its band on every rank that can know them."""
    cost, index = local_record if local_ok else UNUSABLE_RECORD
    gc, gi = select_best_distributed(cost, index, group=group, device=device)
    if fail_inside_selection:
        local_ok = False            # (teb_amd_select_best_distributed returned this rank's error together with the peers' choice)
    if gi < 0:
        return local_ok, -1, None
    owner = owner_of(gi)
    ok, band = broadcast_band(owner, local_bands.get(gi), capacity, local_status=1 if fail_before_broadcast else 0, group=group, device=device)
    return bool(local_ok and ok and not fail_before_broadcast), gi, band
