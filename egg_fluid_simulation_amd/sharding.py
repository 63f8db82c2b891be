"""Multi-GPU sharding of the particle step: one process per GPU, 1-D slabs along x.

The reference is single-process (no distributed backend exists), so this is new
design.  The step only couples particles in adjacent spatial-hash cells
(simulation_handler.lua:1568-1578), and the device path already isolates work in *tiles* of
batches that provably cannot interact.  Sharding therefore works on whole batches:

  * every rank owns the batches whose target lies in its x-slab and steps them with its own
    SimulationHandler -- no particle data crosses ranks while no batch comes near a cut;
  * once per step neighbouring ranks exchange the boxes of their batches that lie within
    `halo_px` of the shared cut (point-to-point isend/irecv, RCCL over xGMI with the "nccl"
    backend: a 1-D chain uses one link per neighbour pair).  These are the ghost records of
    the halo exchange SURVEY.md 8e asks for; at batch granularity they are 4 doubles + an id
    per boundary batch instead of per-particle records;
  * a ghost box within interaction range of a local batch means the two batches could meet in
    this step.  Exact Gauss-Seidel order across a cut cannot be kept in parallel (8e, exactness
    caveat), so that pair must be stepped by ONE rank: ShardedSimulationHandler hands the batch of
    the higher rank over to the lower rank (complete particle state, egg_export_batch /
    egg_import_batch; two messages per neighbour and round, device to device over RCCL) before the step, and hands batches that have strayed deep into another slab
    to that slab's rank.  Every handler lays its particles out in ascending global batch id, so the
    result is bit-identical to one handler holding everything.  BoundaryExchange alone (bench.py)
    raises SlabConflict instead of migrating.

Not covered: the collision budget's exact-mode (a type with so few particles that 0.05 N^2 visits
can bind) needs all particles in one tile and therefore on one rank.  ShardedSimulationHandler.step
watches for it (per-pass visits of the step in flight, summed over ranks against the global budget,
BEFORE the step is committed) and raises EggError instead of stepping on.

Collectives: ONE small all_reduce per step (4 flags: conflict seen / budget suspect / batch strayed out of its
slab's halo / a claim reaches past a whole neighbouring slab) tells every rank whether the launched step may be
committed; everything else on the data path is neighbour point-to-point.  The rare hand-over round exchanges its
plan with two all_gathers of fixed-shape integer tensors (no pickling).
"""
import math

import numpy as np

from .simulation_handler import EggError


class SlabConflict(RuntimeError):
    """A local batch and a neighbour rank's batch are close enough to interact."""


class SlabLayout:
    """x-slabs [cuts[r], cuts[r+1]) for r in range(world)."""

    def __init__(self, cuts):
        self.cuts = [float(c) for c in cuts]
        if any(b <= a for a, b in zip(self.cuts, self.cuts[1:])):
            raise ValueError("slab cuts must ascend")
        self.world = len(self.cuts) - 1

    @classmethod
    def uniform(cls, x_lo, x_hi, world, align=1.0):
        """equal slabs with cuts on multiples of `align` (use the spatial-hash cell size)"""
        cuts = [x_lo + (x_hi - x_lo) * r / world for r in range(world + 1)]
        return cls([np.floor(c / align) * align for c in cuts[:-1]] + [np.ceil(cuts[-1] / align) * align])

    def owner_of(self, x):
        x = np.asarray(x, dtype=np.float64)
        r = np.searchsorted(np.array(self.cuts[1:-1]), x, side="right")
        return r.astype(np.int64)

    def bounds(self, rank):
        return self.cuts[rank], self.cuts[rank + 1]


class BoundaryExchange:
    """Per-step exchange of slab-boundary batch boxes with the left and right neighbour rank.

    The exchanged boxes are the batches' CLAIMS for the upcoming step (SimulationHandler.prepare_step +
    get_bounds): two batches on different ranks are independent iff their claims stay at least one
    spatial-hash cell apart -- the same criterion that separates tiles inside one handler.
    Boxes are [n, 8]: the white claim then the yolk claim; interact_px = (white cell, yolk cell).

    handler      SimulationHandler of this rank, or None when `bounds_fn` is given
    bounds_fn    () -> (ids[n] int64, boxes[n, 4] float64 px) of the local batches (tests inject this)
    group        a torch.distributed module/process group handle (None: world must be 1)
    """

    RECORD = 9  # id, white box (lo_x, lo_y, hi_x, hi_y), yolk box

    def __init__(self, handler, rank, world, slab_lo, slab_hi, group=None, halo_px=64.0, interact_px=(8.0, 12.0),
                 capacity=4096, bounds_fn=None, device=None):
        self.handler, self.rank, self.world = handler, int(rank), int(world)
        self.slab_lo, self.slab_hi = float(slab_lo), float(slab_hi)
        if np.isscalar(interact_px):
            interact_px = (interact_px, interact_px)
        self.halo_px, self.interact_px, self.capacity = float(halo_px), tuple(float(v) for v in interact_px), int(capacity)
        self.dist = group
        self.bounds_fn = bounds_fn or self._handler_bounds
        self.claims_fixed = False
        self.ghosts = {}          # neighbour rank -> (ids, boxes) received in the last exchange
        self.sent = {}            # neighbour rank -> number of boxes sent
        self.bytes_exchanged = 0
        if self.world > 1:
            import torch
            self.torch = torch
            if device is None:
                device = "cuda" if group.get_backend() == "nccl" else "cpu"
            self.device = device
            n = 1 + self.RECORD * self.capacity
            self._send = {r: torch.zeros(n, dtype=torch.float64, device=device) for r in self._neighbours()}
            self._recv = {r: torch.zeros(n, dtype=torch.float64, device=device) for r in self._neighbours()}
            # host images of the messages (pinned when the wire tensors live on the GPU): no per-step
            # allocation, one async copy each way
            pin = str(device).startswith("cuda")
            self._send_host = {r: torch.zeros(n, dtype=torch.float64, pin_memory=pin) for r in self._neighbours()}
            self._recv_host = {r: torch.zeros(n, dtype=torch.float64, pin_memory=pin) for r in self._neighbours()}

    def _neighbours(self):
        return [r for r in (self.rank - 1, self.rank + 1) if 0 <= r < self.world]

    def _handler_bounds(self):
        if not self.claims_fixed:  # a step already launched (step_begin) has fixed this step's claims
            self.handler.prepare_step()
        ids = np.asarray(self.handler.list_ids(), dtype=np.int64)
        boxes, cells = self.handler.get_claims(ids)
        self.interact_px = cells
        return ids, boxes

    def select_boundary(self, ids, boxes, towards):
        """local batches within halo_px of the cut shared with rank `towards`"""
        if towards < self.rank:
            m = np.minimum(boxes[:, 0], boxes[:, 4]) < self.slab_lo + self.halo_px
        else:
            m = np.maximum(boxes[:, 2], boxes[:, 6]) > self.slab_hi - self.halo_px
        return ids[m], boxes[m]

    @staticmethod
    def conflicts(ids, boxes, ghost_ids, ghost_boxes, reach):
        """pairs (local id, ghost id) whose claims are LESS than one hash cell apart in both x and y, for
        the white boxes (columns 0-3, cell reach[0]) or the yolk boxes (columns 4-7, cell reach[1])"""
        if np.isscalar(reach):
            reach = (reach, reach)
        if len(ids) == 0 or len(ghost_ids) == 0:
            return []
        boxes = np.asarray(boxes, dtype=np.float64).reshape(len(ids), -1)
        ghost_boxes = np.asarray(ghost_boxes, dtype=np.float64).reshape(len(ghost_ids), -1)
        if boxes.shape[1] == 4:  # one box per batch: use it for both types
            boxes = np.concatenate([boxes, boxes], axis=1)
        if ghost_boxes.shape[1] == 4:
            ghost_boxes = np.concatenate([ghost_boxes, ghost_boxes], axis=1)
        ids = np.asarray(ids)
        ghost_ids = np.asarray(ghost_ids)
        near = np.zeros((len(ids), len(ghost_ids)), dtype=bool)
        for t in (0, 1):
            b, g, c = boxes[:, None, 4 * t:4 * t + 4], ghost_boxes[None, :, 4 * t:4 * t + 4], reach[t]
            near |= ((b[..., 0] - g[..., 2] < c) & (g[..., 0] - b[..., 2] < c) &
                     (b[..., 1] - g[..., 3] < c) & (g[..., 1] - b[..., 3] < c))
        li, gi = np.nonzero(near.T)[::-1]
        return [(int(ids[a]), int(ghost_ids[b_])) for a, b_ in zip(li, gi)]

    def post(self, claims_fixed=False):
        """First half of the exchange: collect this rank's claims and start the sends / receives.
        claims_fixed: the step was already launched with step_begin (its tiling fixed the claims), so the
        host work of the exchange runs while the kernels do."""
        if self.world == 1:
            return
        self.claims_fixed = claims_fixed
        torch, dist = self.torch, self.dist
        ids, boxes = self.bounds_fn()
        ids = np.asarray(ids, dtype=np.int64)
        boxes = np.asarray(boxes, dtype=np.float64).reshape(len(ids), -1) if len(ids) else np.zeros((0, 8))
        if boxes.shape[1] == 4:
            boxes = np.concatenate([boxes, boxes], axis=1)
        ops = []
        for r in self._neighbours():
            bi, bb = self.select_boundary(ids, boxes, r)
            if len(bi) > self.capacity:
                raise RuntimeError("more than %d boundary batches; raise BoundaryExchange(capacity=...)" % self.capacity)
            rec = self._send_host[r].numpy()
            rec[0] = len(bi)
            body = rec[1:1 + self.RECORD * len(bi)].reshape(len(bi), self.RECORD)
            body[:, 0] = bi
            body[:, 1:] = bb
            if self._send[r].data_ptr() != self._send_host[r].data_ptr():
                self._send[r].copy_(self._send_host[r], non_blocking=True)
            self.sent[r] = len(bi)
            ops.append(dist.P2POp(dist.isend, self._send[r], r))
            ops.append(dist.P2POp(dist.irecv, self._recv[r], r))
        self._pending = dist.batch_isend_irecv(ops)
        self.last_ids, self.last_boxes = ids, boxes

    def finish(self, raise_on_conflict=True):
        """Second half: wait for the neighbours' claims and test them against the local ones.  Returns
        the conflicts as (local id, ghost id, ghost rank); raises SlabConflict for them unless told
        otherwise."""
        if self.world == 1:
            return []
        for req in self._pending:
            req.wait()
        self._pending = []
        ids, boxes = self.last_ids, self.last_boxes
        found = []
        for r in self._neighbours():
            self._recv_host[r].copy_(self._recv[r])
            rec = self._recv_host[r].numpy()
            n = int(rec[0])
            body = rec[1:1 + self.RECORD * n].reshape(n, self.RECORD)
            gids, gboxes = body[:, 0].astype(np.int64), body[:, 1:].copy()
            self.ghosts[r] = (gids, gboxes)
            self.bytes_exchanged += 8 * (2 + self.RECORD * (n + self.sent[r]))
            if n:
                # only local batches that reach as far towards that cut as the ghosts reach into this slab
                # (at least the halo) can meet them
                c = max(self.interact_px)
                if r < self.rank:
                    depth = max(self.halo_px, float(np.max(gboxes[:, [2, 6]])) - self.slab_lo + c)
                    m = np.minimum(boxes[:, 0], boxes[:, 4]) < self.slab_lo + depth
                else:
                    depth = max(self.halo_px, self.slab_hi - float(np.min(gboxes[:, [0, 4]])) + c)
                    m = np.maximum(boxes[:, 2], boxes[:, 6]) > self.slab_hi - depth
                found += [(i, g, r) for i, g in self.conflicts(ids[m], boxes[m], gids, gboxes, self.interact_px)]
        if raise_on_conflict and found:
            raise SlabConflict(
                "rank %d: %d local/ghost batch pairs less than one hash cell apart across a slab cut (first: %s); "
                "use ShardedSimulationHandler to hand such batches over" % (self.rank, len(found), found[:1]))
        return found

    def exchange(self, raise_on_conflict=True):
        """post() + finish()"""
        self.post()
        return self.finish(raise_on_conflict)


class ShardedSimulationHandler:
    """SimulationHandler spread over the ranks of a process group, one x-slab per rank.

    SPMD: every rank makes the same add / set_target_position / update calls with the same
    arguments; batch ids are global.  `make_handler` builds the local device handler (tests inject
    their own).  Results equal a single handler's bit for bit (tests/test_gpu_sharded.py).
    """

    STATE_FIELDS = 9

    def __init__(self, layout, rank, group, make_handler, halo_px=64.0, interact_px=(8.0, 12.0), device=None):
        import torch
        self.torch, self.dist = torch, group
        self.layout, self.rank, self.world = layout, int(rank), layout.world
        self.local = make_handler()
        self.owner = {}       # global id -> rank
        self.local_id = {}    # global id -> id in self.local (batches this rank owns)
        self.global_id = {}   # local id -> global id
        self.radii = {}       # global id -> (white_radius, yolk_radius)
        self.next_gid = 1
        self.migrations = 0
        self.bytes_handed_over = 0
        self._committed_visits = [0, 0]  # most pairs one pass of the last committed step visited, per type
        self._elapsed = 0.0
        self.interpolation_alpha = 0.0
        self._budget_stale = True
        self._step_args = (1 / 60, 2, 3)
        lo, hi = layout.bounds(rank)
        self.exchange = BoundaryExchange(None, rank, self.world, lo, hi, group=group, halo_px=halo_px,
                                         interact_px=interact_px, bounds_fn=self._bounds, device=device)
        self.device = self.exchange.device if self.world > 1 else "cpu"

    # ------------------------------------------------------------------ API
    def add(self, x, y, white_radius=50.0, yolk_radius=15.0):
        gid = self.next_gid
        self.next_gid += 1
        r = int(self.layout.owner_of([x])[0])
        self.owner[gid] = r
        self.radii[gid] = (white_radius, yolk_radius)
        if r == self.rank:
            lid = int(self.local.add_many_keyed([x], [y], [gid], white_radius, yolk_radius)[0])
            self.local_id[gid] = lid
            self.global_id[lid] = gid
        self._budget_stale = True  # summed over the ranks before the next step
        return gid

    def set_target_position(self, gid, x, y):
        if self.owner[gid] == self.rank:
            self.local.set_target_position(self.local_id[gid], x, y)

    def update(self, delta, step_delta=None, n_substeps=None, n_collision_steps=None):
        """The reference's fixed-step accumulator (simulation_handler.lua:199-216) around the exchanging step():
        every `_step` gets its own claim exchange (the claims of one exchange cover one step only)."""
        step_delta = 1 / 60 if step_delta is None else step_delta
        n_substeps = 2 if n_substeps is None else int(math.ceil(n_substeps))
        n_collision_steps = 3 if n_collision_steps is None else int(math.ceil(n_collision_steps))
        if not (step_delta > 0) or n_substeps < 1 or n_collision_steps < 1:
            raise EggError("[ERROR] In SimulationHandler.update: invalid step_delta / n_substeps / n_collision_steps")
        self._elapsed = self._elapsed + delta
        n_steps = 0
        max_n_steps = max(4.0, 4 * math.ceil((1 / 60) / step_delta))
        while self._elapsed >= step_delta:
            self.step(step_delta, n_substeps, n_collision_steps)
            self._elapsed = self._elapsed - step_delta
            n_steps += 1
            if n_steps > max_n_steps:  # death-spiral guard, L:208-213
                self._elapsed = 0
                break
        self.interpolation_alpha = min(max(self._elapsed / step_delta, 0.0), 1.0)
        return n_steps

    def step(self, delta=1 / 60, n_substeps=2, n_collision_steps=3):
        """One `_step` on every rank with the neighbour exchange hidden behind the kernels: post the
        claims, launch the local step, then look at the neighbours' claims; only if some pair of batches
        on different ranks could interact is the launched step discarded, the batches handed over and
        the step run again."""
        self._step_args = (delta, n_substeps, n_collision_steps)
        if self.world == 1:
            self.local.step(delta, n_substeps, n_collision_steps)
            return 0
        self._sync_budget()
        self.local.step_begin(delta, n_substeps, n_collision_steps)  # fixes this step's claims, launches the kernels
        self.exchange.post(claims_fixed=True)
        conflicts = self.exchange.finish(raise_on_conflict=False)
        ids, boxes = self.exchange.last_ids, self.exchange.last_boxes
        lo, hi = self.exchange.slab_lo, self.exchange.slab_hi
        halo = self.exchange.halo_px
        stray = wide = False
        if len(ids):
            bl, bh = np.minimum(boxes[:, 0], boxes[:, 4]), np.maximum(boxes[:, 2], boxes[:, 6])
            # strayed: wholly beyond the halo of this slab (it belongs to the neighbour now)
            stray = bool(np.any((bh < lo - halo) & (self.rank > 0)) or np.any((bl > hi + halo) & (self.rank + 1 < self.world)))
            # wide: a claim reaches past the whole neighbouring slab, where only the rank after next could see it
            left_far = self.layout.cuts[self.rank - 1] if self.rank > 0 else -np.inf
            right_far = self.layout.cuts[self.rank + 2] if self.rank + 2 <= self.world else np.inf
            wide = bool(np.any(bl < left_far + halo) and self.rank > 1) or bool(np.any(bh > right_far - halo) and self.rank + 2 < self.world)
        # The budget guard (simulation_handler.lua:1657-1658): the reference counts the visits of ALL particles
        # against 0.05 N^2, a rank only sees its own.  While every rank stays below budget / world the global count
        # cannot bind; a rank above it triggers the exact sum -- of THIS step, which is still uncommitted.
        visits, budget = self.local.step_peek_visits()
        # (the peek sees the FIRST attempt of the step in flight; step_end may re-run it after a failed claim check, and
        # the step after a hand-over is not peeked at all: what those COMMITTED steps visited is folded into the next
        # step's test -- a binding budget is then refused one step late instead of never)
        visits = [max(int(v), int(c)) for v, c in zip(visits, self._committed_visits)]
        suspect = any(v * self.world > max(1.0, math.ceil(b)) for v, b in zip(visits, budget))
        flag = self.torch.tensor([1.0 if conflicts else 0.0, 1.0 if suspect else 0.0, 1.0 if stray else 0.0,
                                  1.0 if wide else 0.0], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        flags = flag.tolist()
        if flags[1] != 0.0:
            tot = self.torch.tensor([float(v) for v in visits], dtype=self.torch.float64, device=self.device)
            self.dist.all_reduce(tot, op=self.dist.ReduceOp.SUM)
            for which, (v, b) in enumerate(zip(tot.tolist(), budget)):
                if v > max(1.0, math.ceil(b)):
                    self.local.step_end(False)  # nothing of the suspect step is kept
                    raise EggError("collision budget may bind across ranks (type %d: up to %d visits in a pass, "
                                   "budget %.2f): exact-budget mode needs all particles of the type on one rank"
                                   % (which, int(v), b))
        if flags[3] != 0.0:
            self.local.step_end(False)
            raise EggError("a batch's claim for this step reaches past a whole neighbouring slab; slabs must be wider "
                           "than the distance a batch travels in one step plus the halo")
        if flags[0] == 0.0:
            self.local.step_end(True)
            self._note_committed()
            # a batch that left its slab's halo without meeting anything is handed to the slab it is in before the
            # next step, so that every batch is always known to the ranks on both sides of it
            return self.rebalance() if flags[2] != 0.0 else 0
        self.local.step_end(False)
        moved = self.rebalance()
        self.local.step(delta, n_substeps, n_collision_steps)
        self._note_committed()
        return moved

    def _note_committed(self):
        st = self.local.stats()
        self._committed_visits = [int(v) for v in st["max_pass_visits"]]

    def positions(self):
        """{global id: (x, y)} of every batch, gathered on all ranks"""
        mine = {g: self.local.get_position(l) for g, l in self.local_id.items()}
        if self.world == 1:
            return mine
        rec = np.array([[g, p[0], p[1]] for g, p in sorted(mine.items())], dtype=np.float64).reshape(-1, 3)
        merged = {}
        for part in self._all_gather_rows(rec):
            for g, x, y in part:
                merged[int(g)] = (float(x), float(y))
        return merged

    def _all_gather_rows(self, rows):
        """all_gather of a [n, k] float64 array with a different n on every rank: counts first, then the rows padded to
        the largest count (two fixed-shape tensor collectives, nothing is pickled)"""
        torch, dist = self.torch, self.dist
        rows = np.asarray(rows, dtype=np.float64)
        k = rows.shape[1]
        cnt = torch.tensor([float(rows.shape[0])], dtype=torch.float64, device=self.device)
        counts = [torch.zeros(1, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        dist.all_gather(counts, cnt)
        counts = [int(c.item()) for c in counts]
        m = max(1, max(counts))
        buf = torch.zeros(m * k, dtype=torch.float64, device=self.device)
        if rows.shape[0]:
            buf[:rows.size] = torch.from_numpy(rows.reshape(-1)).to(self.device)
        parts = [torch.zeros(m * k, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        dist.all_gather(parts, buf)
        return [p.cpu().numpy()[:c * k].reshape(c, k) for p, c in zip(parts, counts)]

    def particles(self, which):
        """{global id: (x[n], y[n])} of this rank's batches"""
        x, y, b = self.local.download(which, "x"), self.local.download(which, "y"), self.local.download(which, "batch_id")
        return {self.global_id[int(l)]: (x[b == l], y[b == l]) for l in np.unique(b)}

    # ------------------------------------------------------------ internals
    def _bounds(self):
        if not self.exchange.claims_fixed:
            self.local.prepare_step(*self._step_args)
        gids = np.array(sorted(self.local_id), dtype=np.int64)
        if len(gids) == 0:
            return gids, np.zeros((0, 8))
        boxes, cells = self.local.get_claims([self.local_id[int(g)] for g in gids])
        self.exchange.interact_px = cells
        return gids, boxes

    def _sync_budget(self):
        """the budget 0.05 N^2 counts the particles of ALL ranks (simulation_handler.lua:1752-1753): the sum of what the
        handlers actually hold (hand-overs do not change it; adds and removes mark it stale)"""
        if self.world == 1 or not self._budget_stale:
            return
        from . import _ffi
        nw, ny = self.local.get_n_particles()
        tot = self.torch.tensor([float(nw), float(ny)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(tot, op=self.dist.ReduceOp.SUM)
        nw, ny = (int(v) for v in tot.tolist())
        self.local.set_option(_ffi.OPT_BUDGET_PARTICLES_WHITE, nw)
        self.local.set_option(_ffi.OPT_BUDGET_PARTICLES_YOLK, ny)
        self._budget_stale = False

    HEAD = 7  # gid, target x, y, white radius, yolk radius, white particles, yolk particles

    def _send_batches(self, gids, to):
        """hands the batches `gids` to rank `to`: ONE header message and ONE payload message for all of them.  With the
        wire tensors on the GPU ("nccl" = RCCL) the particle state goes device to device: egg_export_batch writes it
        straight into the tensor RCCL sends, nothing passes through host memory."""
        torch, dist = self.torch, self.dist
        gids = list(gids)
        if not gids:
            return
        on_gpu = str(self.device).startswith("cuda") and hasattr(self.local, "export_batch_to")
        counts = [self.local.get_n_particles(self.local_id[g]) for g in gids] if on_gpu else None
        heads, parts = [], []
        if on_gpu:
            total = sum(self.STATE_FIELDS * (nw + ny) for nw, ny in counts)
            payload = torch.empty(total, dtype=torch.float64, device=self.device)
            off = 0
            for g, (nw, ny) in zip(gids, counts):
                w0, y0 = off, off + self.STATE_FIELDS * nw
                info = self.local.export_batch_to(self.local_id[g], payload.data_ptr() + 8 * w0, payload.data_ptr() + 8 * y0)
                heads += [g, info["target_x"], info["target_y"], info["white_radius"], info["yolk_radius"], nw, ny]
                off = y0 + self.STATE_FIELDS * ny
        else:
            for g in gids:
                info, ws, ys = self.local.export_batch(self.local_id[g])
                heads += [g, info["target_x"], info["target_y"], info["white_radius"], info["yolk_radius"], info["n_white"], info["n_yolk"]]
                parts += [np.asarray(ws, dtype=np.float64).reshape(-1), np.asarray(ys, dtype=np.float64).reshape(-1)]
            payload = torch.from_numpy(np.concatenate(parts)).to(self.device)
        head = torch.tensor(heads, dtype=torch.float64).to(self.device)
        dist.send(head, to)
        dist.send(payload, to)
        self.bytes_handed_over += 8 * (head.numel() + payload.numel())
        for g in gids:
            lid = self.local_id.pop(g)
            del self.global_id[lid]
            self.local.remove(lid)

    def _recv_batches(self, n, frm):
        """receives the `n` batches rank `frm` hands over in this round (the plan told every rank how many)"""
        torch, dist = self.torch, self.dist
        if n == 0:
            return []
        head = torch.zeros(self.HEAD * n, dtype=torch.float64, device=self.device)
        dist.recv(head, frm)
        rows = head.cpu().numpy().reshape(n, self.HEAD)
        total = int(sum(self.STATE_FIELDS * (int(r[5]) + int(r[6])) for r in rows))
        payload = torch.zeros(total, dtype=torch.float64, device=self.device)
        dist.recv(payload, frm)
        on_gpu = str(self.device).startswith("cuda") and hasattr(self.local, "import_batch_from")
        host = None if on_gpu else payload.cpu().numpy()
        got, off = [], 0
        for r in rows:
            gid, nw, ny = int(r[0]), int(r[5]), int(r[6])
            info = dict(key=gid, target_x=float(r[1]), target_y=float(r[2]), white_radius=float(r[3]), yolk_radius=float(r[4]),
                        n_white=nw, n_yolk=ny)
            w0, y0 = off, off + self.STATE_FIELDS * nw
            if on_gpu:
                lid = self.local.import_batch_from(info, payload.data_ptr() + 8 * w0, payload.data_ptr() + 8 * y0)
            else:
                lid = self.local.import_batch(info, host[w0:y0].reshape(self.STATE_FIELDS, nw),
                                              host[y0:y0 + self.STATE_FIELDS * ny].reshape(self.STATE_FIELDS, ny))
            off = y0 + self.STATE_FIELDS * ny
            self.local_id[gid] = lid
            self.global_id[lid] = gid
            got.append(gid)
        return got

    def rebalance(self, max_rounds=None):
        """Before a step: exchange boundary boxes and hand batches over until no local batch is within
        interaction range of another rank's batch.  Conflicts move the higher rank's batch down;
        batches that strayed past the halo of their slab move to the slab they are in."""
        if self.world == 1:
            return 0
        torch, dist = self.torch, self.dist
        moved_total = 0
        for _ in range(max_rounds or 2 * self.world + 2):
            conflicts = self.exchange.exchange(raise_on_conflict=False)
            ids, boxes = self.exchange.last_ids, self.exchange.last_boxes
            lo, hi = self.exchange.slab_lo, self.exchange.slab_hi
            to_left, to_right, want_right = set(), set(), set()
            # batches move as whole ISLANDS (local batches chained by claims less than a cell apart): one member going
            # alone would meet its island-mates across the cut in the next round and come straight back
            island = {int(g): int(g) for g in ids}

            def find(g):
                while island[g] != g:
                    island[g] = island[island[g]]
                    g = island[g]
                return g

            for a, b in self.exchange.conflicts(ids, boxes, ids, boxes, self.exchange.interact_px):
                if a != b:
                    ra, rb = find(int(a)), find(int(b))
                    if ra != rb:
                        island[max(ra, rb)] = min(ra, rb)
            members = {}
            for g in island:
                members.setdefault(find(g), []).append(g)
            for lid_g, ghost, ghost_rank in conflicts:
                if ghost_rank < self.rank:
                    to_left.update(members[find(int(lid_g))])   # the lower rank steps the pair
                else:
                    want_right.add(int(ghost))  # its owner may not see my batch: ask for it
            in_conflict = {find(int(c[0])) for c in conflicts}
            box_of = {int(g): b for g, b in zip(ids, boxes)}
            for root, gs in members.items():
                if root in in_conflict:
                    continue
                # strayed: the whole island lies beyond the halo of this slab -> it belongs to the slab it is in
                if self.rank > 0 and all(max(box_of[g][2], box_of[g][6]) < lo - self.exchange.halo_px for g in gs):
                    to_left.update(gs)
                elif self.rank + 1 < self.world and all(min(box_of[g][0], box_of[g][4]) > hi + self.exchange.halo_px for g in gs):
                    to_right.update(gs)
            # everyone learns every plan: how many batches arrive from whom, and the new owner table
            rows = [[0.0, g] for g in sorted(to_left)] + [[1.0, g] for g in sorted(to_right)] + [[2.0, g] for g in sorted(want_right)]
            parts = self._all_gather_rows(np.array(rows, dtype=np.float64).reshape(-1, 2))
            gathered = [([int(g) for k, g in part if k == 0.0], [int(g) for k, g in part if k == 1.0],
                         [int(g) for k, g in part if k == 2.0]) for part in parts]
            plan = []
            for r, (l, rr, _w) in enumerate(gathered):
                wanted = set(gathered[r - 1][2]) if r > 0 else set()
                l = sorted(set(l) | {g for g in wanted if self.owner.get(g) == r})
                plan.append((l, [g for g in rr if g not in l]))
            to_left, to_right = set(plan[self.rank][0]), set(plan[self.rank][1])
            n_moves = sum(len(l) + len(r) for l, r in plan)
            if n_moves == 0:
                # nothing left to move anywhere: every rank decides TOGETHER whether that is a clean end (a rank that
                # raised alone would leave the others hanging in their next collective)
                bad = torch.tensor([1.0 if conflicts else 0.0], dtype=torch.float64, device=self.device)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if bad.item() != 0.0:
                    raise SlabConflict("rank %d: unresolved cross-slab pairs %s" % (self.rank, conflicts[:3]))
                return moved_total
            # even ranks send first, odd ranks receive first: neighbour pairs never both block in send
            for phase in (0, 1):
                if self.rank % 2 == phase:
                    self._send_batches(sorted(to_left), self.rank - 1)
                    self._send_batches(sorted(to_right), self.rank + 1)
                else:
                    if self.rank + 1 < self.world:
                        self._recv_batches(len(plan[self.rank + 1][0]), self.rank + 1)
                    if self.rank > 0:
                        self._recv_batches(len(plan[self.rank - 1][1]), self.rank - 1)
            for r, (l, rr) in enumerate(plan):
                for g in l:
                    self.owner[g] = r - 1
                for g in rr:
                    self.owner[g] = r + 1
            moved_total += n_moves
            self.migrations += n_moves
        raise SlabConflict("rank %d: batch hand-over did not settle" % self.rank)
