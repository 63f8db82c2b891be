"""Developer: per-phase cycle breakdown of the step kernel (tile 0), diagnostic build."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EGGSIM_LIB"] = os.path.join(ROOT, "egg_fluid_simulation_amd", "libeggsim_prof.so")
sys.path.insert(0, ROOT)
from egg_fluid_simulation_amd import SimulationHandler, _ffi
import numpy as np
names = ["load", "pre+follow", "hash", "count", "fill", "budget", "transpose", "dag", "post", "writeback"]
def run(nb, steps=20):
    side = int(np.ceil(np.sqrt(abs(nb))))
    if nb < 0:  # -k: k batches overlapping in one island
        nb = -nb
        xs = np.array([10.0 + 50 * k for k in range(nb)]); ys = np.array([10.0] * nb)
    else:
        xs = np.array([100 + 160.0 * (k % side) for k in range(nb)]); ys = np.array([100 + 160.0 * (k // side) for k in range(nb)])
    h = SimulationHandler(); h.set_option(_ffi.OPT_TIMING, 1)
    if 'EGG_SPREAD' in os.environ: h.set_option(_ffi.OPT_THREADS_PER_PARTICLE, int(os.environ['EGG_SPREAD']))
    h.add_many(xs, ys, 50, 15)
    S = int(os.environ.get('EGG_S', '2')); Cc = int(os.environ.get('EGG_C', '3'))
    for _ in range(5): h.step(1 / 60, S, Cc)
    L = _ffi.load(); L.egg_prof_reset()
    kms = 0
    for _ in range(steps): h.step(1 / 60, S, Cc); kms += h.stats()["last_step_kernel_ms"]
    buf = (C.c_ulonglong * 32)(); L.egg_prof_read(buf)
    calls = buf[11]
    tot = sum(buf[k] for k in range(10)) + sum(buf[k] for k in range(12, 26))
    print("S=%d C=%d" % (S, Cc), "batches=%d kernel %.3f ms/step; tile-0 kernels=%d total ticks/kernel=%.0f rounds/kernel=%.0f" % (nb, kms / steps, calls, tot / calls, buf[10] / calls))
    fine = {12: "hash: clear", 13: "hash: keys+count", 14: "hash: cell scan", 15: "hash: scatter", 16: "transpose: scan", 17: "transpose: scatter", 18: "offsets scan"}
    for k, nm in fine.items(): print("   %-18s %10.0f ticks/kernel  %5.1f%%" % (nm, buf[k] / calls, 100.0 * buf[k] / tot))
    print("   (the coarse rows below now exclude the fine rows above: hash = in-cell rank only, transpose = rank pass only, fill = offsets->fill)")
    for k in range(10): print("   %-10s %10.0f ticks/kernel  %5.1f%%" % (names[k], buf[k] / calls, 100.0 * buf[k] / tot))
if __name__ == "__main__":
    for nb in [int(a) for a in sys.argv[1:]] or [1]: run(nb)
