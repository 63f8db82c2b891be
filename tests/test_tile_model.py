"""The GPU tile algorithm (visit lists from cell adjacency + budget cut + dependency-DAG pair
scheduler), stated in Python in tests/tile_model.py, against the CPU oracle -- bit for bit.
This is the CPU-side proof that the kernel's scheduling rules reproduce the reference's
sequential Gauss-Seidel order, including the stale first pass of every later sub-step (Q3)
and the collision budget (Q2).  No GPU needed."""
import math

import numpy as np
import pytest

from conftest import circle_target
from tile_model import TileModel


def _run(oracle_mod, centers, steps, S=2, C=3):
    o = oracle_mod.Oracle()
    ids = [o.add(cx, cy, 50, 15) for cx, cy in centers]
    tms = []
    for which, cfg, R in ((0, oracle_mod.DEFAULT_WHITE, 50.0), (1, oracle_mod.DEFAULT_YOLK, 15.0)):
        x, y = o.positions(which)
        tms.append(TileModel(cfg, x, y, o.field(which, "inv_mass"), o.field(which, "radius"),
                             o.field(which, "batch_id").astype(int) - 1, [tuple(c) for c in centers],
                             [2 * math.sqrt(R)] * len(centers)))
    rounds = []
    for k in range(steps):
        for i, c in zip(ids, centers):
            t = circle_target(c, k)
            o.set_target_position(i, *t)
            for tm in tms:
                tm.targets[i - 1] = t
        o.step(1 / 60, S, C)
        ps = o.pass_stats()
        for which, tm in enumerate(tms):
            tm.step(1 / 60, S, C)
            xo, yo = o.positions(which)
            assert np.array_equal(np.array(tm.st["x"]), xo) and np.array_equal(np.array(tm.st["y"]), yo), (k, which)
            assert [s["n_visited"] for s in ps if s["which"] == which] == [l[0] for l in tm.log]
        rounds.append([l[2] for l in tms[0].log])
    return rounds


def test_single_batch_with_budget_cut_yolk(oracle_mod):
    rounds = _run(oracle_mod, [(400, 300)], 6)
    # the DAG is much shallower than the ~720 sequential pair solves of a pass
    assert max(max(r) for r in rounds) < 260


def test_overlapping_batches_negative_coordinates(oracle_mod):
    _run(oracle_mod, [(0, 0), (30, 10), (-20, 40)], 4)


@pytest.mark.parametrize("S,C", [(1, 2), (2, 1), (3, 2), (3, 1), (5, 1)])
def test_substep_and_pass_variants(oracle_mod, S, C):
    _run(oracle_mod, [(10, 10), (60, 10)], 3, S, C)
