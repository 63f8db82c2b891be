"""Developer timing: steps/s and step-kernel time at a given batch count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egg_fluid_simulation_amd import SimulationHandler, _ffi
import numpy as np

def run(nb, steps=50, warm=10, pack=0):
    side = int(np.ceil(np.sqrt(nb)))
    xs = np.array([100 + 160.0 * (k % side) for k in range(nb)]); ys = np.array([100 + 160.0 * (k // side) for k in range(nb)])
    h = SimulationHandler()
    h.set_option(_ffi.OPT_TIMING, 1)
    if pack: h.set_option(_ffi.OPT_TILE_TARGET_PARTICLES, pack)
    sp = int(os.environ.get('EGG_SPREAD', '1'))
    if sp != 1: h.set_option(_ffi.OPT_THREADS_PER_PARTICLE, sp)
    if 'EGG_SLEEP' in os.environ: h.set_option(_ffi.OPT_SPIN_SLEEP, int(os.environ['EGG_SLEEP']))
    h.add_many(xs, ys, 50, 15)
    for _ in range(warm): h.step()
    t0 = time.perf_counter(); kms = 0.0
    for _ in range(steps):
        h.step(); kms += h.stats()["last_step_kernel_ms"]
    dt = time.perf_counter() - t0
    st = h.stats()
    print("batches=%d pack=%d  wall %.3f ms/step  kernel %.3f ms/step  tiles=%s retiles=%d redo=%d pair_solves/step=%.0f"
          % (nb, pack, 1e3 * dt / steps, kms / steps, st["n_tiles"], st["retiles"], st["redo_steps"], st["pair_solves"] / st["steps"]))
    sys.stdout.flush()

if __name__ == "__main__":
    pack = int(os.environ.get('EGG_PACK', '0'))
    for nb in [int(a) for a in sys.argv[1:]] or [1, 16, 256, 1024, 4096]:
        run(nb, pack=pack)
