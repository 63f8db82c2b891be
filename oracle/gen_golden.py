"""Generates tests/golden/*.npz from the pure-Python transliteration (oracle/reference_model.py).

The reference itself cannot be run in this pipeline (no Lua interpreter), so these vectors are
NOT outputs of the reference; they pin the C oracle and the device path to the independently
written line-by-line transliteration ("parity unpinned", see oracle/eggsim_oracle.h).

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import reference_model as rm  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: (centers, moving target?, n_steps, snapshot steps, substeps, collision steps)
    "cfg1_static": ([(400.0, 300.0)], False, 100, [1, 2, 10, 100], 2, 3),
    "cfg1_moving": ([(400.0, 300.0)], True, 100, [1, 2, 10, 100], 2, 3),
    "cfg1_origin": ([(0.0, 0.0)], True, 50, [1, 10, 50], 2, 3),
    "four_batches": ([(0.0, 0.0), (30.0, 10.0), (-20.0, 40.0), (200.0, 200.0)], True, 20, [1, 5, 20], 2, 3),
    "substeps_3_2": ([(10.0, 10.0), (60.0, 10.0)], True, 12, [1, 12], 3, 2),
    "substeps_2_1": ([(10.0, 10.0), (20.0, 20.0)], True, 12, [1, 12], 2, 1),
}


def target(center, k):
    """gate-B trajectory: 100 px circle, one revolution per 100 steps"""
    return (center[0] + 100 * math.cos(2 * math.pi * k / 100), center[1] + 100 * math.sin(2 * math.pi * k / 100))


def run_case(name):
    centers, moving, n_steps, snaps, S, C = CASES[name]
    m = rm.ReferenceModel()
    ids = [m.add(cx, cy, 50, 15) for cx, cy in centers]
    out = {"centers": np.array(centers), "moving": np.array(moving), "n_steps": np.array(n_steps),
           "snap_steps": np.array(snaps), "substeps": np.array(S), "collision_steps": np.array(C)}
    out["init_white"] = np.array([m.field(0, f) for f in (rm.X, rm.Y, rm.MASS_T, rm.INV_MASS, rm.RADIUS)])
    out["init_yolk"] = np.array([m.field(1, f) for f in (rm.X, rm.Y, rm.MASS_T, rm.INV_MASS, rm.RADIUS)])
    visits = []
    for k in range(n_steps):
        if moving:
            for i, c in zip(ids, centers):
                m.set_target_position(i, *target(c, k))
        n = m.update(1 / 60, 1 / 60, S, C)
        assert n == 1
        visits.append([p[3] for p in m.pass_log])
        if k + 1 in snaps:
            for w, tag in ((0, "white"), (1, "yolk")):
                out["%s_step%d" % (tag, k + 1)] = np.array(
                    [m.field(w, rm.X), m.field(w, rm.Y), m.field(w, rm.VX), m.field(w, rm.VY)])
            out["centroid_step%d" % (k + 1)] = np.array([m.get_position(i) for i in ids])
    out["visits"] = np.array(visits, dtype=np.int64)  # [step][pass: white, yolk alternating]
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in CASES:
        data = run_case(name)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
        print("wrote", name, {k: v.shape for k, v in data.items() if k.startswith("white_step")})


if __name__ == "__main__":
    main()
