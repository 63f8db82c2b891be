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
