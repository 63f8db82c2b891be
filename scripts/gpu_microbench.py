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
names = {2: "1 fma chain", 3: "2 fma chains", 4: "4 fma chains", 5: "rcp chain", 6: "rsq chain",
         7: "cmp+branch+fma", 8: "LDS store->load", 9: "f32 fma chain", 10: "i32 mad chain",
         11: "f64 cmp + int add", 12: "f64 add chain", 13: "f64 mul chain", 14: "s_nop 15 (16 core cycles)"}
import time
for rep in range(2):
    for mode in [14] + list(range(2, 14)) + [14]:
        best = None
        for k in range(4):
            c = C.c_ulonglong()
            t0 = time.perf_counter()
            L.egg_microbench(mode, 20000, 64, C.byref(c))
            dt = time.perf_counter() - t0
            v = c.value / 20000.0 / 32
            best = v if best is None else min(best, v)
        print("mode=%2d (%s): %.2f ticks per instruction; last launch %.0f ticks/us wall" % (mode, names[mode], best, c.value / (dt * 1e6)))
print("several waves of one workgroup running the f64 fma chain (mode 2) side by side:")
for waves in (1, 2, 4, 8, 12, 16):
    best = None
    for k in range(3):
        c = C.c_ulonglong()
        L.egg_microbench(100 * waves + 2, 20000, 64, C.byref(c))
        v = c.value / 20000.0 / 32
        best = v if best is None else min(best, v)
    print("  %2d waves: %.2f ticks per instruction per wave -> %.2f ticks per instruction per CU" % (waves, best, best / waves))
