"""Multi-GPU host logic on CPU: slab layout and the per-step boundary-box exchange between
neighbour ranks, run as 2 real processes over the gloo backend (the GPU path uses the same code
over nccl = RCCL).  The per-rank solver needs a GPU, so bounds are injected."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_slab_layout():
    from egg_fluid_simulation_amd.sharding import SlabLayout
    lay = SlabLayout.uniform(0.0, 1000.0, 4, align=8.0)
    assert lay.world == 4 and lay.cuts[0] == 0.0 and lay.cuts[-1] == 1000.0
    assert all(c % 8.0 == 0 for c in lay.cuts[:-1])  # cuts sit on spatial-hash cell boundaries
    assert list(lay.owner_of([0.0, 247.9, 248.0, 999.0])) == [0, 0, 1, 3]
    assert lay.bounds(1) == (248.0, 496.0)
    with pytest.raises(ValueError):
        SlabLayout([0, 10, 5])


def test_conflict_detection():
    from egg_fluid_simulation_amd.sharding import BoundaryExchange
    ids = np.array([1, 2])
    boxes = np.array([[0.0, 0.0, 100.0, 100.0], [300.0, 0.0, 400.0, 100.0]])
    ghosts = np.array([[130.0, 50.0, 200.0, 90.0], [1000.0, 0.0, 1100.0, 10.0]])
    assert BoundaryExchange.conflicts(ids, boxes, np.array([7, 8]), ghosts, 48.0) == [(1, 7)]
    assert BoundaryExchange.conflicts(ids, boxes, np.array([7, 8]), ghosts, 10.0) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, scenario, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from egg_fluid_simulation_amd.sharding import BoundaryExchange, SlabConflict, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = SlabLayout.uniform(0.0, 2000.0, world, align=8.0)
        lo, hi = lay.bounds(rank)
        # 4 batches per rank in a row inside the slab; ids are globally unique
        xs = np.array([lo + 120 + 240.0 * k for k in range(4)])
        if scenario == "conflict" and rank == 0:
            xs[-1] = hi - 30  # pushed against the cut
        if scenario == "conflict" and rank == 1:
            xs[0] = lo + 40
        ids = np.arange(4) + 100 * rank
        boxes = np.stack([xs - 56, np.full(4, 44.0), xs + 56, np.full(4, 156.0)], axis=1)
        ex = BoundaryExchange(None, rank, world, lo, hi, group=dist, bounds_fn=lambda: (ids, boxes), halo_px=64.0,
                              interact_px=48.0, capacity=16)
        try:
            ex.exchange()
            outcome = "ok"
        except SlabConflict as e:
            outcome = "conflict"
        ghosts = {r: (g[0].tolist(), g[1].tolist()) for r, g in ex.ghosts.items()}
        q.put((rank, outcome, ghosts, dict(ex.sent)))
    except Exception:  # surface the traceback in the parent instead of a queue timeout
        import traceback
        q.put((rank, "error: " + traceback.format_exc(), {}, {}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["separated", "conflict"])
def test_boundary_exchange_two_ranks_gloo(scenario):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, outcome, ghosts, sent = q.get(timeout=120)
        assert not outcome.startswith("error"), outcome
        res[rank] = (outcome, ghosts, sent)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if scenario == "separated":
        # batches sit >= 64 px inside their slabs: nothing is near the cut, nothing is sent
        assert res[0][0] == res[1][0] == "ok"
        assert res[0][2] == {1: 0} and res[1][2] == {0: 0}
        assert res[0][1][1][0] == [] and res[1][1][0][0] == []
    else:
        # each rank learns about the other's boundary batch and both flag the same pair
        assert res[0][0] == res[1][0] == "conflict"
        assert res[0][1][1][0] == [100] and res[1][1][0][0] == [3]
        assert res[0][1][1][1][0][0] == pytest.approx(1000.0 + 40 - 56)  # white box lo_x of the ghost


# ---------------------------------------------------------------------------------------------------------------
# ShardedSimulationHandler's protocol on the CPU, four ranks: a stand-in for the device handler (each batch is a box
# that walks towards its target at a fixed speed; "particle state" is its centre) behind the real exchange /
# hand-over code over gloo.  A batch column is driven across TWO cuts, and the scenario is arranged so that the
# middle rank receives from BOTH neighbours in one hand-over round.

class _FakeHandler:
    """the part of SimulationHandler's surface sharding.py uses; no GPU"""
    SPEED = 25.0

    def __init__(self):
        self.b = {}       # local id -> dict(key, x, y, tx, ty)
        self.next_id = 1
        self.flight = None
        self.imports = []  # (round marker, key) for the test
        self.options = {}

    def add_many_keyed(self, xs, ys, keys, wr=None, yr=None):
        ids = []
        for x, y, k in zip(xs, ys, keys):
            self.b[self.next_id] = dict(key=int(k), x=float(x), y=float(y), tx=float(x), ty=float(y))
            ids.append(self.next_id)
            self.next_id += 1
        return np.array(ids)

    def set_target_position(self, i, x, y):
        self.b[i]["tx"], self.b[i]["ty"] = float(x), float(y)

    def _advance(self):
        new = {}
        for i, v in self.b.items():
            dx, dy = v["tx"] - v["x"], v["ty"] - v["y"]
            d = (dx * dx + dy * dy) ** 0.5
            s = min(1.0, self.SPEED / d) if d > 0 else 0.0
            new[i] = (v["x"] + s * dx, v["y"] + s * dy)
        return new

    def step_begin(self, *a):
        assert self.flight is None
        self.flight = self._advance()

    def step_end(self, commit=True):
        assert self.flight is not None
        if commit:
            for i, (x, y) in self.flight.items():
                self.b[i]["x"], self.b[i]["y"] = x, y
        self.flight = None

    def step(self, *a):
        self.step_begin()
        self.step_end(True)

    def step_peek_visits(self):
        return [0, 0], [1e9, 1e9]

    def prepare_step(self, *a):
        pass

    def get_claims(self, ids):
        # the claim covers where the batch is and where it will be after the step, +- 56 px
        nxt = self._advance()
        out = []
        for i in ids:
            v, (nx, ny) = self.b[int(i)], nxt[int(i)]
            box = [min(v["x"], nx) - 56, min(v["y"], ny) - 56, max(v["x"], nx) + 56, max(v["y"], ny) + 56]
            out.append(box + box)
        return np.array(out, dtype=np.float64).reshape(len(ids), 8), (8.0, 12.0)

    def export_batch(self, i):
        v = self.b[i]
        info = dict(key=v["key"], target_x=v["tx"], target_y=v["ty"], white_radius=50.0, yolk_radius=15.0, n_white=1, n_yolk=1)
        st = np.zeros((9, 1))
        st[0, 0], st[1, 0] = v["x"], v["y"]
        return info, st, st.copy()

    def import_batch(self, info, ws, ys):
        i = self.next_id
        self.next_id += 1
        self.b[i] = dict(key=int(info["key"]), x=float(ws[0, 0]), y=float(ws[1, 0]), tx=info["target_x"], ty=info["target_y"])
        self.imports.append(int(info["key"]))
        return i

    def remove(self, i):
        del self.b[i]

    def get_position(self, i):
        return (self.b[i]["x"], self.b[i]["y"])

    def get_n_particles(self):
        return (len(self.b), len(self.b))

    def set_option(self, k, v):
        self.options[k] = v

    def stats(self):
        return dict(max_pass_visits=[0, 0], budget=[1e9, 1e9])


def _scenario4():
    """cuts at 500 / 1000 / 1500.  Batch 1 (slab 0) runs right through slab 1 into slab 2; batch 2 (slab 2) runs left
    into slab 1 where it meets batch 3 (slab 1, standing still).  Batch 1's target is set so that it has strayed out of
    slab 0's halo exactly when 2 and 3 first come within reach: rank 1 then receives batch 1 from rank 0 (stray) and
    batch 2 from rank 2 (conflict: the lower rank steps the pair) in one hand-over round.  Batches 4-7 stand still."""
    start = {1: (380.0, 100.0), 2: (1180.0, 400.0), 3: (760.0, 400.0), 4: (100.0, 700.0), 5: (1300.0, 700.0),
             6: (1800.0, 100.0), 7: (1700.0, 700.0)}
    goal = {1: (1400.0, 100.0), 2: (880.0, 400.0)}
    return start, goal


def _worker4(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, goal = _scenario4()
        sh = ShardedSimulationHandler(SlabLayout([0.0, 500.0, 1000.0, 1500.0, 2000.0]), rank, dist, _FakeHandler, device="cpu")
        gids = [sh.add(*start[g]) for g in sorted(start)]
        assert gids == sorted(start)
        rounds = []  # per step: keys this rank imported
        for k in range(40):
            for g, t in goal.items():
                sh.set_target_position(g, *t)
            before = len(sh.local.imports)
            if k % 3 == 2:
                assert sh.update(1 / 60) == 1
            else:
                sh.step(1 / 60)
            rounds.append(sh.local.imports[before:])
        q.put((rank, "ok", sh.positions(), dict(sh.owner), sorted(sh.local_id), rounds, sh.migrations))
    except Exception:
        import traceback
        q.put((rank, "error: " + traceback.format_exc(), None, None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_four_ranks_hand_over_from_both_sides_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        out = q.get(timeout=180)
        assert out[1] == "ok", out[1]
        res[out[0]] = out
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the same dynamics in one process: every batch walks towards its target on its own
    start, goal = _scenario4()
    ref = _FakeHandler()
    ids = ref.add_many_keyed([start[g][0] for g in sorted(start)], [start[g][1] for g in sorted(start)], sorted(start))
    for k in range(40):
        for g, t in goal.items():
            ref.set_target_position(int(ids[g - 1]), *t)
        ref.step()
    want = {g: ref.get_position(int(ids[g - 1])) for g in sorted(start)}
    for r in range(4):
        assert res[r][2] == want, r                      # every rank gathers the same, correct positions
        assert res[r][3] == res[0][3]                    # and the same owner table
    held = sorted(g for r in range(4) for g in res[r][4])
    assert held == sorted(start)                          # no batch lost or duplicated
    owner = res[0][3]
    assert owner[1] == 2 and owner[2] == 1 and owner[3] == 1  # batch 1 crossed two cuts; 2 joined 3 on the lower rank
    both = [rd for rd in res[1][5] if 1 in rd and 2 in rd]
    assert both, "rank 1 must have received from both neighbours in one hand-over round: %s" % res[1][5]
    assert res[0][6] == res[3][6] >= 3                   # everyone counts the same migrations


# ---------------------------------------------------------------------------------------------------------------
# Eight ranks: BASELINE config 5's layout (a square grid of batches at 160 px pitch, cut into 8 x-slabs of equal
# column count) at reduced size -- 16 columns x 6 rows, two columns per slab -- with every target moving on the
# gate-B circle (radius 100 px: every batch of a slab's edge column crosses into the neighbour's halo and back, many of
# them past the cut), plus one batch that runs from the first slab to the last through all seven cuts.

def _scenario8():
    cols, rows, pitch = 16, 6, 160.0
    start = {}
    for r in range(rows):
        for c in range(cols):
            start[1 + r * cols + c] = (100.0 + pitch * c, 100.0 + pitch * r)
    runner = len(start) + 1
    start[runner] = (60.0, 100.0 + pitch * rows)  # below the grid, in slab 0
    cuts = [20.0 + 2 * pitch * k for k in range(9)]  # between column pairs: 20, 340, ... 2580
    return start, runner, cuts, pitch * rows + 100.0


def _targets8(start, runner, runner_y, k):
    import math
    ang = 2 * math.pi * k / 40
    out = {g: (x + 100.0 * math.cos(ang) - 100.0, y + 100.0 * math.sin(ang)) for g, (x, y) in start.items() if g != runner}
    out[runner] = (2540.0, runner_y)
    return out


def _worker8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    torch.set_num_threads(1)
    import torch.distributed as dist
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, runner, cuts, runner_y = _scenario8()
        sh = ShardedSimulationHandler(SlabLayout(cuts), rank, dist, _FakeHandler, device="cpu")
        for g in sorted(start):
            assert sh.add(*start[g]) == g
        for k in range(120):
            for g, t in _targets8(start, runner, runner_y, k).items():
                sh.set_target_position(g, *t)
            sh.step(1 / 60)
        q.put((rank, "ok", sh.positions(), dict(sh.owner), sorted(sh.local_id), sh.migrations, sh.bytes_handed_over))
    except Exception:
        import traceback
        q.put((rank, "error: " + traceback.format_exc(), None, None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_config5_slab_layout_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        out = q.get(timeout=300)
        assert out[1] == "ok", out[1]
        res[out[0]] = out
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    start, runner, cuts, runner_y = _scenario8()
    ref = _FakeHandler()
    order = sorted(start)
    ids = ref.add_many_keyed([start[g][0] for g in order], [start[g][1] for g in order], order)
    for k in range(120):
        for g, t in _targets8(start, runner, runner_y, k).items():
            ref.set_target_position(int(ids[g - 1]), *t)
        ref.step()
    want = {g: ref.get_position(int(ids[g - 1])) for g in order}
    for r in range(8):
        assert res[r][2] == want, r                      # every rank gathers the same, correct positions
        assert res[r][3] == res[0][3]                    # and the same owner table
        assert res[r][5] == res[0][5]                    # and counts the same hand-overs
    held = sorted(g for r in range(8) for g in res[r][4])
    assert held == order                                  # no batch lost or duplicated
    assert res[0][3][runner] == 7                         # the runner crossed all seven cuts
    assert res[0][5] >= 10                                # ... and batches of the circling edge columns were handed over too
    assert sum(res[r][6] for r in range(8)) > 0
