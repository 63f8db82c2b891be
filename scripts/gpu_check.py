"""Developer check: device path vs CPU oracle on small cases, with timings.  Run on the GPU box:
    python scripts/gpu_check.py"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egg_fluid_simulation_amd import SimulationHandler, WHITE, YOLK  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def compare(name, centers, steps, moving=True, S=2, C=3, verbose=True):
    h = SimulationHandler()
    o = Oracle()
    ids = [h.add(cx, cy, 50, 15) for cx, cy in centers]
    for cx, cy in centers:
        o.add(cx, cy, 50, 15)
    worst = 0.0
    first_bad = None
    t_gpu = 0.0
    for k in range(steps):
        if moving:
            for i, (cx, cy) in zip(ids, centers):
                tx = cx + 100 * math.cos(2 * math.pi * k / 100)
                ty = cy + 100 * math.sin(2 * math.pi * k / 100)
                h.set_target_position(i, tx, ty)
                o.set_target_position(i, tx, ty)
        t0 = time.perf_counter()
        h.update(1 / 60, 1 / 60, S, C)
        t_gpu += time.perf_counter() - t0
        o.update(1 / 60, 1 / 60, S, C)
        for w in (WHITE, YOLK):
            xo, yo = o.positions(w)
            d = max(np.abs(h.download(w, "x") - xo).max(), np.abs(h.download(w, "y") - yo).max())
            worst = max(worst, d)
            if d != 0 and first_bad is None:
                first_bad = (k, w, d)
    px = max(abs(a - b) for i in ids for a, b in zip(h.get_position(i), o.get_position(i)))
    st = h.stats()
    print("%-28s steps=%d max|dx|=%.3e first_bad=%s centroid_diff=%.3e gpu_ms/step=%.3f pair_solves=%d (oracle %d) "
          "tiles=%s retiles=%d redo=%d single=%s" % (name, steps, worst, first_bad, px, 1e3 * t_gpu / steps,
                                                     st["pair_solves"], o.total_visited, st["n_tiles"],
                                                     st["retiles"], st["redo_steps"], st["single_tile"]))
    sys.stdout.flush()
    return worst


if __name__ == "__main__":
    compare("cfg1 static", [(400, 300)], 100, moving=False)
    compare("cfg1 moving", [(400, 300)], 100)
    compare("origin moving", [(0, 0)], 60)
    compare("4 overlapping", [(0, 0), (30, 10), (-20, 40), (200, 200)], 40)
    compare("S=1,C=2", [(10, 10)], 20, S=1, C=2)
    compare("S=2,C=1", [(10, 10), (20, 20)], 20, S=2, C=1)
    compare("S=3,C=2", [(10, 10), (60, 10)], 20, S=3, C=2)
    grid = [(100 + 160 * i, 100 + 160 * j) for j in range(4) for i in range(4)]
    compare("16 grid static", grid, 30, moving=False)
    compare("16 grid moving", grid, 30)
    grid = [(100 + 160 * i, 100 + 160 * j) for j in range(16) for i in range(16)]
    compare("256 grid static", grid, 10, moving=False)
