import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EGGSIM_LIB"] = os.path.join(ROOT, "egg_fluid_simulation_amd", "libeggsim_prof.so")
sys.path.insert(0, ROOT)
from egg_fluid_simulation_amd import SimulationHandler, _ffi
h = SimulationHandler(); L = _ffi.load()
for mode in (0, 1):
    for lanes in (1, 8, 64):
        c = C.c_ulonglong()
        L.egg_microbench(mode, 2000, lanes, C.byref(c))
        L.egg_microbench(mode, 2000, lanes, C.byref(c))
        print("mode=%d active_lanes=%d: %.1f cycles per projection" % (mode, lanes, c.value / 2000.0))
