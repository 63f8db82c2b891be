import math
import os
import sys

import numpy as np
import pytest

try:  # torch brings its own HIP runtime: it has to be in the process BEFORE libeggsim.so pulls in the system's copy,
    import torch  # noqa: F401  -- or torch.cuda finds no GPU afterwards (the device-to-device hand-over test uses CUDA tensors)
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["cfg1_static", "cfg1_moving", "cfg1_origin", "four_batches", "substeps_3_2", "substeps_2_1"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def circle_target(center, k):
    """gate-B trajectory of oracle/gen_golden.py"""
    return (center[0] + 100 * math.cos(2 * math.pi * k / 100), center[1] + 100 * math.sin(2 * math.pi * k / 100))


def replay_golden(g, sim, positions, on_snapshot):
    """Drives `sim` (oracle or device handler: add / set_target_position / update) through the
    golden case `g`; calls on_snapshot(step, tag, array[4, n]) with sim's x, y, vx, vy."""
    centers = [tuple(c) for c in g["centers"]]
    S, C = int(g["substeps"]), int(g["collision_steps"])
    ids = [sim.add(cx, cy, 50, 15) for cx, cy in centers]
    snaps = set(int(s) for s in g["snap_steps"])
    for k in range(int(g["n_steps"])):
        if bool(g["moving"]):
            for i, c in zip(ids, centers):
                sim.set_target_position(i, *circle_target(c, k))
        assert sim.update(1 / 60, 1 / 60, S, C) == 1
        if k + 1 in snaps:
            for w, tag in ((0, "white"), (1, "yolk")):
                on_snapshot(k + 1, tag, positions(sim, w))
    return ids


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as om
    om.build()
    return om
