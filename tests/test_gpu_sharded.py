"""Two and four ranks (one process and one handler each on the same GPU, gloo for the exchange -- the GPU box
has one card; the code path is the one nccl/RCCL runs over xGMI) against ONE CPU oracle holding all
batches: columns of batches are driven across the slab cuts into each other, so batches are handed
over between ranks mid-run (a middle rank receives from both sides).  Positions must match the
single-handler result bit for bit.  Also BASELINE config 4's slab layout (4 x-slabs of batch columns) at
reduced size."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_STEPS = 70


def _scenario():
    """10 batches in two columns either side of the cut at x = 1000, 300 px apart inside a column (so
    only opposite batches meet, as pairs); targets make the columns swap sides.  10 batches keep the
    yolk collision budget from binding (it binds for <= 9 batches)."""
    centers = [(760.0, 150.0 + 300.0 * k) for k in range(5)] + [(1240.0, 150.0 + 300.0 * k) for k in range(5)]

    def target(gid, step):
        cx, cy = centers[gid - 1]
        t = min(1.0, step / 40.0)
        return (cx + (480.0 if cx < 1000 else -480.0) * t, cy + 10.0 * t)

    return centers, target


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from egg_fluid_simulation_amd import SimulationHandler
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        centers, target = _scenario()
        layout = SlabLayout([0.0, 1000.0, 2000.0])
        sh = ShardedSimulationHandler(layout, rank, dist, lambda: SimulationHandler(device=0), device="cpu")
        gids = [sh.add(x, y, 50, 15) for x, y in centers]
        owners0 = dict(sh.owner)
        for k in range(N_STEPS):
            for g in gids:
                sh.set_target_position(g, *target(g, k))
            if k % 2 == 0:
                sh.step(1 / 60)  # exchange overlapped with the kernels, re-run after a hand-over
            else:
                assert sh.update(1 / 60) == 1  # exchange and hand-over first, then the step
        white = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(0).items()}
        yolk = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(1).items()}
        pos = sh.positions()
        q.put((rank, white, yolk, pos, sh.migrations, owners0, dict(sh.owner)))
    finally:
        dist.destroy_process_group()


def _budget_worker(rank, world, port, q):
    """one batch per rank: 30 yolk particles globally, budget 0.05 * 30^2 = 45 visits per pass, which the
    two blobs together exceed -> the reference's early return would fire -> must be refused, not stepped"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from egg_fluid_simulation_amd import EggError, SimulationHandler
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = ShardedSimulationHandler(SlabLayout([0.0, 1000.0, 2000.0]), rank, dist,
                                      lambda: SimulationHandler(device=0), device="cpu")
        sh.add(500.0, 500.0, 50, 15)
        sh.add(1500.0, 500.0, 50, 15)
        raised_at = None
        for k in range(4):
            try:
                sh.step(1 / 60)
            except EggError as e:
                raised_at = (k, str(e))
                break
        q.put((rank, raised_at, sh.local.stats()["max_pass_visits"], sh.local.stats()["budget"]))
    finally:
        dist.destroy_process_group()


def _run_two(worker, world=2):
    import queue
    import time

    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    deadline = time.time() + 300
    while len(res) < world and time.time() < deadline:
        try:
            out = q.get(timeout=2)
            res[out[0]] = out
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(20)
        if p.is_alive():
            p.kill()  # the exact child started above
    assert len(res) == world and all(p.exitcode == 0 for p in procs), "a rank failed: see its traceback above"
    return res


def test_binding_budget_across_ranks_is_refused():
    res = _run_two(_budget_worker)
    for r in (0, 1):
        _, raised_at, visits, budget = res[r]
        # the visits of the step IN FLIGHT are summed over the ranks before anything is committed: refused at once
        assert raised_at is not None and raised_at[0] == 0, (raised_at, visits, budget)
        # every rank prices the budget on the GLOBAL particle count: 0.05 * 30^2
        assert "budget 45.00" in raised_at[1], raised_at[1]
        assert budget == [0.0, 0.0]  # nothing was committed


def test_two_ranks_with_hand_over_match_single_handler(oracle_mod):
    res = _run_two(_worker)

    centers, target = _scenario()
    o = oracle_mod.Oracle()
    gids = [o.add(x, y, 50, 15) for x, y in centers]
    for k in range(N_STEPS):
        for g in gids:
            o.set_target_position(g, *target(g, k))
        o.update(1 / 60)
    for which in (0, 1):
        x, y, b = o.field(which, "x"), o.field(which, "y"), o.field(which, "batch_id")
        seen = set()
        for r in (0, 1):
            for g, (gx, gy) in res[r][1 + which].items():
                assert g not in seen
                seen.add(g)
                assert np.array_equal(np.array(gx), x[b == g]) and np.array_equal(np.array(gy), y[b == g]), (which, g)
        assert seen == set(gids)
    for g in gids:
        assert res[0][3][g] == o.get_position(g) and res[1][3][g] == o.get_position(g)
    assert res[0][4] > 0, "the scenario must force hand-overs"
    assert res[0][5] != res[0][6], "ownership must have changed"
    assert res[0][6] == res[1][6], "both ranks agree on who owns what"


# ------------------------------------------------------------------------------------------------ four ranks

N4_STEPS = 60


def _scenario4():
    """cuts at 600 / 1200 / 1800.  Twelve batches (the yolk budget stays slack): a column of three in slab 0 is driven
    right across slab 1 into slab 2, a column of three in slab 2 is driven left into slab 1, where three batches stand;
    three more stand in slab 3.  Rank 1 receives batches from both neighbours, batches cross two cuts, islands of
    batches from three slabs form and dissolve."""
    rows = [150.0, 450.0, 750.0]
    centers = ([(420.0, y) for y in rows] + [(900.0, y + 20.0) for y in rows] + [(1420.0, y - 10.0) for y in rows] +
               [(2100.0, y) for y in rows])

    def target(gid, step):
        cx, cy = centers[gid - 1]
        t = min(1.0, step / 45.0)
        if gid <= 3:
            return (cx + 1000.0 * t, cy + 15.0 * t)   # slab 0 -> slab 2
        if 7 <= gid <= 9:
            return (cx - 470.0 * t, cy)               # slab 2 -> slab 1
        return (cx, cy)

    return centers, target


def _worker4(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from egg_fluid_simulation_amd import SimulationHandler
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        centers, target = _scenario4()
        layout = SlabLayout([0.0, 600.0, 1200.0, 1800.0, 2400.0])
        sh = ShardedSimulationHandler(layout, rank, dist, lambda: SimulationHandler(device=0), device="cpu")
        gids = [sh.add(x, y, 50, 15) for x, y in centers]
        owners0 = dict(sh.owner)
        received_from = set()
        for k in range(N4_STEPS):
            for g in gids:
                sh.set_target_position(g, *target(g, k))
            before = dict(sh.owner)
            if k % 2 == 0:
                sh.step(1 / 60)
            else:
                assert sh.update(1 / 60) == 1
            for g, r in sh.owner.items():
                if r == rank and before[g] != rank:
                    received_from.add(before[g])
        white = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(0).items()}
        yolk = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(1).items()}
        q.put((rank, white, yolk, sh.positions(), sh.migrations, owners0, dict(sh.owner), sorted(received_from)))
    finally:
        dist.destroy_process_group()


def _check_against_oracle(oracle_mod, res, world, centers, target, n_steps):
    o = oracle_mod.Oracle()
    gids = [o.add(x, y, 50, 15) for x, y in centers]
    for k in range(n_steps):
        for g in gids:
            o.set_target_position(g, *target(g, k))
        o.update(1 / 60)
    for which in (0, 1):
        x, y, b = o.field(which, "x"), o.field(which, "y"), o.field(which, "batch_id")
        seen = set()
        for r in range(world):
            for g, (gx, gy) in res[r][1 + which].items():
                assert g not in seen
                seen.add(g)
                assert np.array_equal(np.array(gx), x[b == g]) and np.array_equal(np.array(gy), y[b == g]), (which, g)
        assert seen == set(gids)
    for g in gids:
        for r in range(world):
            assert res[r][3][g] == o.get_position(g)
    return gids


def test_four_ranks_with_hand_over_across_two_cuts_match_single_handler(oracle_mod):
    res = _run_two(_worker4, world=4)
    centers, target = _scenario4()
    _check_against_oracle(oracle_mod, res, 4, centers, target, N4_STEPS)
    assert res[0][4] >= 6, "the scenario must force hand-overs"
    for r in range(1, 4):
        assert res[r][6] == res[0][6], "all ranks agree on who owns what"
    owner = res[0][6]
    assert all(owner[g] in (1, 2) for g in (1, 2, 3)) and res[0][5][1] == 0  # the slab-0 column left slab 0 ...
    assert set(res[1][7]) == {0, 2}, "rank 1 must have received batches from both neighbours: %s" % res[1][7]


def _worker_cfg4(rank, world, port, q):
    """BASELINE config 4's layout at reduced size: a 16 x 6 grid of batches at 160 px pitch cut into 4 x-slabs of four
    columns; every target moves on a small circle (no batch comes near a cut: the exchange runs every step and finds
    nothing to hand over)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import math

    import torch.distributed as dist
    from egg_fluid_simulation_amd import SimulationHandler
    from egg_fluid_simulation_amd.sharding import ShardedSimulationHandler, SlabLayout
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        centers = [(100.0 + 160.0 * c, 100.0 + 160.0 * r) for r in range(6) for c in range(16)]
        layout = SlabLayout([20.0 + 640.0 * k for k in range(5)])
        sh = ShardedSimulationHandler(layout, rank, dist, lambda: SimulationHandler(device=0), device="cpu")
        gids = [sh.add(x, y, 50, 15) for x, y in centers]
        for k in range(8):
            dx, dy = 20.0 * math.cos(0.5 * k), 20.0 * math.sin(0.5 * k)
            for g, (x, y) in zip(gids, centers):
                sh.set_target_position(g, x + dx, y + dy)
            sh.step(1 / 60)
        white = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(0).items()}
        yolk = {g: (x.tolist(), y.tolist()) for g, (x, y) in sh.particles(1).items()}
        q.put((rank, white, yolk, sh.positions(), sh.migrations, len(sh.local_id), sh.exchange.bytes_exchanged))
    finally:
        dist.destroy_process_group()


def test_config4_slab_layout_reduced_four_ranks(oracle_mod):
    import math
    res = _run_two(_worker_cfg4, world=4)
    centers = [(100.0 + 160.0 * c, 100.0 + 160.0 * r) for r in range(6) for c in range(16)]

    def target(gid, k):
        x, y = centers[gid - 1]
        return (x + 20.0 * math.cos(0.5 * k), y + 20.0 * math.sin(0.5 * k))

    _check_against_oracle(oracle_mod, res, 4, centers, target, 8)
    for r in range(4):
        assert res[r][4] == 0 and res[r][5] == 24  # four columns of six per slab, nothing handed over
        assert res[r][6] > 0                         # but the neighbours did talk every step
