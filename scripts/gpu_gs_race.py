"""Experiment: large island through the global-memory-state kernel, run repeatedly (EGGSIM_LIB picks the build)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from egg_fluid_simulation_amd import SimulationHandler, EggError
k = np.arange(16)
xs, ys = 500.0 + 95.0 * (k % 4), 500.0 + 95.0 * (k // 4)
ref = None
for trial in range(6):
    h = SimulationHandler()
    h.add_many(xs, ys, 50, 15)
    try:
        for _ in range(3):
            h.step(1 / 60, 2, 3)
        x = h.download(0, "x")
        if ref is None:
            ref = x
        print("trial", trial, "ok, identical to first:", bool(np.array_equal(x, ref)), flush=True)
    except EggError as e:
        print("trial", trial, "FAILED:", e, flush=True)
