"""Developer: config 2 with every target moving on a circle (the interactive use of the reference: blobs chase
the cursor) -- steps/s including the per-step target upload and any re-tiling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from egg_fluid_simulation_amd import SimulationHandler, _ffi
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0  # px per step
h = SimulationHandler()
xs, ys, side = bench.grid_positions(nb)
ids = h.add_many(xs, ys, 50, 15)
h.set_option(_ffi.OPT_TIMING, 1)
def run(n, k0):
    for k in range(k0, k0 + n):
        a = 0.05 * k
        h.set_target_positions(ids, xs + 20 * speed * np.cos(a), ys + 20 * speed * np.sin(a))
        h.step(1 / 60, 2, 3)
run(50, 0)
s0 = h.stats(); t0 = time.perf_counter()
run(300, 50)
dt = time.perf_counter() - t0; s1 = h.stats()
print("yolk kernel %.3f ms" % (s1["kernel_ms_sum"][1] / max(1, s1["timed_steps"])))
print("host ms/step: prepare %.3f launch %.3f wait %.3f" % tuple((s1["host_ms"][i] - s0["host_ms"][i]) / 300 for i in range(3)))
print("%d batches, targets moving %.1f px/step: %.1f steps/s, %.3f ms/step, white kernel %.3f ms, retiles %d, redo %d of 300 steps" % (
    nb, speed, 300 / dt, 1e3 * dt / 300, s1["kernel_ms_sum"][0] / max(1, s1["timed_steps"]), s1["retiles"] - s0["retiles"], s1["redo_steps"] - s0["redo_steps"]))
