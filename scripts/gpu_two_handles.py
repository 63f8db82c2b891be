"""Developer experiment: does the chip have room for two packed pipelines side by side?  Config 3 split over P handles
(each on its own streams), stepped from P threads, against one handle holding everything."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from egg_fluid_simulation_amd import SimulationHandler, _ffi

nb, overlap, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = 60


def make(n):
    xs, ys, side = bench.grid_positions(n, overlap=overlap)
    h = SimulationHandler()
    h.set_option(_ffi.OPT_PACKED, 1)
    h.add_many(xs, ys, 50, 15)
    for _ in range(10):
        h.step(1 / 60, 2, 3)
    return h


def run(hs):
    def work(h):
        for _ in range(steps):
            h.step(1 / 60, 2, 3)
    ts = [threading.Thread(target=work, args=(h,)) for h in hs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return (time.perf_counter() - t0) / steps * 1e3


one = make(nb)
print("%d batches x%d in ONE handle: %.3f ms/step" % (nb, overlap, run([one])))
del one
parts = [make(nb // P) for _ in range(P)]
print("%d handles of %d batches side by side: %.3f ms/step for all" % (P, nb // P, run(parts)))
print("one of those alone: %.3f ms/step" % run(parts[:1]))
